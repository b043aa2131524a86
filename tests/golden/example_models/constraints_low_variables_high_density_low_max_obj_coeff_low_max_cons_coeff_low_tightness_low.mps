NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_90_0
 L  R_90_1
 L  R_90_2
 L  R_90_3
COLUMNS
    x_0       OBJROW     -1.           R_90_0    3.          
    x_1       OBJROW     -2.           R_90_3    7.          
RHS
    RHS       R_90_0    2.             R_90_1    2.          
    RHS       R_90_2    2.             R_90_3    2.          
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
