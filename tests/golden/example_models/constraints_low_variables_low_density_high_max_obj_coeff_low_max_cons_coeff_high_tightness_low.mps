NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_66_0
 L  R_66_1
COLUMNS
    x_0       OBJROW     -1.           R_66_0    22.         
    x_0       R_66_1    86.         
    x_1       OBJROW     -2.           R_66_1    28.         
RHS
    RHS       R_66_0    99.            R_66_1    81.         
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
