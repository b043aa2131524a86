NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_146_0
 L  R_146_1
 L  R_146_2
 L  R_146_3
COLUMNS
    x_0       OBJROW     -8.           R_146_0   22.         
    x_0       R_146_1   86.            R_146_3   28.         
    x_1       OBJROW     -12.          R_146_0   75.         
    x_1       R_146_1   56.            R_146_2   93.         
RHS
    RHS       R_146_0   96.            R_146_1   91.         
    RHS       R_146_2   97.            R_146_3   94.         
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
ENDATA
