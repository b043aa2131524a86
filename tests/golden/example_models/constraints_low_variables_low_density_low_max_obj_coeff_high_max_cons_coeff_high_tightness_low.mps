NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_50_0
 L  R_50_1
COLUMNS
    x_0       OBJROW     -8.           R_50_0    22.         
    x_1       OBJROW     -12.       
RHS
    RHS       R_50_0    24.            R_50_1    21.         
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
ENDATA
