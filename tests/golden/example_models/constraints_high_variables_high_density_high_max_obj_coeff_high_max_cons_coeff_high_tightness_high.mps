NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_278_0
 L  R_278_1
 L  R_278_2
 L  R_278_3
COLUMNS
    x_0       OBJROW     -8.           R_278_1   86.         
    x_0       R_278_3   28.         
    x_1       OBJROW     -12.          R_278_0   75.         
    x_1       R_278_1   56.            R_278_2   93.         
    x_2       OBJROW     -11.          R_278_0   48.         
    x_2       R_278_1   57.            R_278_2   5.          
    x_3       OBJROW     -47.          R_278_0   41.         
    x_3       R_278_2   68.            R_278_3   23.         
RHS
    RHS       R_278_0   100.           R_278_1   99.         
    RHS       R_278_2   46.            R_278_3   85.         
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
 UI BOUND     x_2       100.        
 UI BOUND     x_3       100.        
ENDATA
