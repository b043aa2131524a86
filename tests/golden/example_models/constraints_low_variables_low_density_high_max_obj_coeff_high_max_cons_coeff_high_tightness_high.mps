NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_86_0
 L  R_86_1
COLUMNS
    x_0       OBJROW     -8.           R_86_0    22.         
    x_0       R_86_1    86.         
    x_1       OBJROW     -12.          R_86_1    28.         
RHS
    RHS       R_86_0    97.            R_86_1    24.         
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
ENDATA
