NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_106_0
 L  R_106_1
 L  R_106_2
 L  R_106_3
COLUMNS
    x_0       OBJROW     -8.           R_106_0   3.          
    x_1       OBJROW     -12.          R_106_3   7.          
RHS
    RHS       R_106_0   2.             R_106_1   2.          
    RHS       R_106_2   2.             R_106_3   2.          
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
ENDATA
