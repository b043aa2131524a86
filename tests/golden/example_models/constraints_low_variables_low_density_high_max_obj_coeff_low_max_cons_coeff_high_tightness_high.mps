NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_70_0
 L  R_70_1
COLUMNS
    x_0       OBJROW     -1.           R_70_0    22.         
    x_0       R_70_1    86.         
    x_1       OBJROW     -2.           R_70_1    28.         
RHS
    RHS       R_70_0    97.            R_70_1    24.         
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
