NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_138_0
 L  R_138_1
 L  R_138_2
 L  R_138_3
COLUMNS
    x_0       OBJROW     -8.           R_138_0   3.          
    x_0       R_138_1   5.             R_138_2   4.          
    x_0       R_138_3   10.         
    x_1       OBJROW     -12.          R_138_0   7.          
    x_1       R_138_1   9.             R_138_3   8.          
RHS
    RHS       R_138_0   10.            R_138_1   9.          
    RHS       R_138_2   8.             R_138_3   8.          
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
ENDATA
