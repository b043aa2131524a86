NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_34_0
 L  R_34_1
COLUMNS
    x_0       OBJROW     -1.           R_34_0    22.         
    x_1       OBJROW     -2.        
RHS
    RHS       R_34_0    24.            R_34_1    21.         
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
