NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_142_0
 L  R_142_1
 L  R_142_2
 L  R_142_3
COLUMNS
    x_0       OBJROW     -8.           R_142_0   3.          
    x_0       R_142_1   5.             R_142_2   4.          
    x_0       R_142_3   10.         
    x_1       OBJROW     -12.          R_142_0   7.          
    x_1       R_142_1   9.             R_142_3   8.          
RHS
    RHS       R_142_0   10.            R_142_1   6.          
    RHS       R_142_2   2.             R_142_3   2.          
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
ENDATA
