NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_210_0
 L  R_210_1
COLUMNS
    x_0       OBJROW     -8.           R_210_1   86.         
    x_1       OBJROW     -12.          R_210_1   28.         
    x_2       OBJROW     -11.          R_210_0   75.         
    x_2       R_210_1   56.         
    x_3       OBJROW     -47.          R_210_0   93.         
RHS
    RHS       R_210_0   192.           R_210_1   183.        
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
 UI BOUND     x_2       100.        
 UI BOUND     x_3       100.        
ENDATA
