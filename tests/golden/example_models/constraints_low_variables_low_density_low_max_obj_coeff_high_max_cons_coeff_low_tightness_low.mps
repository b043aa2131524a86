NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_42_0
 L  R_42_1
COLUMNS
    x_0       OBJROW     -8.           R_42_0    3.          
    x_1       OBJROW     -12.       
RHS
    RHS       R_42_0    2.             R_42_1    2.          
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
ENDATA
