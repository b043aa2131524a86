NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_770_0
 L  R_770_1
COLUMNS
    x_0       OBJROW     -14.          R_770_0   11.         
    x_1       OBJROW     -14.          R_770_1   39.         
    x_2       OBJROW     -1.           R_770_0   3.          
    x_3       OBJROW     -2.           R_770_0   28.         
RHS
    RHS       R_770_0   34.            R_770_1   19.         
BOUNDS
 UI BOUND     x_0       14.         
 UI BOUND     x_1       14.         
 UI BOUND     x_2       14.         
 UI BOUND     x_3       14.         
ENDATA
