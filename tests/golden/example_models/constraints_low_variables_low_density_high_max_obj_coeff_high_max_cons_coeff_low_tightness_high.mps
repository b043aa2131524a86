NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_78_0
 L  R_78_1
COLUMNS
    x_0       OBJROW     -8.           R_78_0    3.          
    x_0       R_78_1    5.          
    x_1       OBJROW     -12.          R_78_0    4.          
    x_1       R_78_1    10.         
RHS
    RHS       R_78_0    4.             R_78_1    8.          
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
ENDATA
