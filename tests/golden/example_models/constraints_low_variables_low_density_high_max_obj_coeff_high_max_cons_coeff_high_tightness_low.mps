NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_82_0
 L  R_82_1
COLUMNS
    x_0       OBJROW     -8.           R_82_0    22.         
    x_0       R_82_1    86.         
    x_1       OBJROW     -12.          R_82_1    28.         
RHS
    RHS       R_82_0    99.            R_82_1    81.         
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
ENDATA
