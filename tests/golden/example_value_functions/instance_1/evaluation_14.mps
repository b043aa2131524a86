NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_270_0
 L  R_270_1
COLUMNS
    x_0       OBJROW     -2.           R_270_1   20.         
    x_1       OBJROW     -3.           R_270_0   14.         
    x_1       R_270_1   38.         
    x_2       OBJROW     -3.           R_270_0   28.         
    x_2       R_270_1   33.         
    x_3       OBJROW     -12.          R_270_1   29.         
RHS
    RHS       R_270_0   82.            R_270_1   66.         
BOUNDS
 UI BOUND     x_0       26.         
 UI BOUND     x_1       26.         
 UI BOUND     x_2       26.         
 UI BOUND     x_3       26.         
ENDATA
