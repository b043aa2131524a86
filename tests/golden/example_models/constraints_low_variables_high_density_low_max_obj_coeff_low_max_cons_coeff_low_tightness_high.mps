NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_94_0
 L  R_94_1
 L  R_94_2
 L  R_94_3
COLUMNS
    x_0       OBJROW     -1.           R_94_0    3.          
    x_1       OBJROW     -2.           R_94_3    7.          
RHS
    RHS       R_94_0    2.             R_94_1    1.          
    RHS       R_94_2    2.             R_94_3    2.          
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
