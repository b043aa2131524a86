NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_26_0
 L  R_26_1
COLUMNS
    x_0       OBJROW     -1.           R_26_0    3.          
    x_1       OBJROW     -2.        
RHS
    RHS       R_26_0    2.             R_26_1    2.          
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
