NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_118_0
 L  R_118_1
 L  R_118_2
 L  R_118_3
COLUMNS
    x_0       OBJROW     -8.           R_118_0   22.         
    x_1       OBJROW     -12.          R_118_3   56.         
RHS
    RHS       R_118_0   25.            R_118_1   17.         
    RHS       R_118_2   21.            R_118_3   16.         
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
ENDATA
