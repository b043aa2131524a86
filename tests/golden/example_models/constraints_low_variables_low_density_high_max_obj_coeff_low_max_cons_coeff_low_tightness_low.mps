NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_58_0
 L  R_58_1
COLUMNS
    x_0       OBJROW     -1.           R_58_0    3.          
    x_0       R_58_1    5.          
    x_1       OBJROW     -2.           R_58_0    4.          
    x_1       R_58_1    10.         
RHS
    RHS       R_58_0    10.            R_58_1    8.          
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
