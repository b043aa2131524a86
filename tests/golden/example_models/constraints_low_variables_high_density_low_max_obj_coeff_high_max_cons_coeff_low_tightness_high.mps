NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_110_0
 L  R_110_1
 L  R_110_2
 L  R_110_3
COLUMNS
    x_0       OBJROW     -8.           R_110_0   3.          
    x_1       OBJROW     -12.          R_110_3   7.          
RHS
    RHS       R_110_0   2.             R_110_1   1.          
    RHS       R_110_2   2.             R_110_3   2.          
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
ENDATA
