NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_30_0
 L  R_30_1
COLUMNS
    x_0       OBJROW     -1.           R_30_0    3.          
    x_1       OBJROW     -2.        
RHS
    RHS       R_30_0    2.          
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
