NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_330_0
 L  R_330_1
COLUMNS
    x_0       OBJROW     -2.           R_330_1   20.         
    x_1       OBJROW     -3.           R_330_0   14.         
    x_1       R_330_1   38.         
    x_2       OBJROW     -3.           R_330_0   28.         
    x_2       R_330_1   33.         
    x_3       OBJROW     -12.          R_330_1   29.         
RHS
    RHS       R_330_0   38.            R_330_1   9.          
BOUNDS
 UI BOUND     x_0       26.         
 UI BOUND     x_1       26.         
 UI BOUND     x_2       26.         
 UI BOUND     x_3       26.         
ENDATA
