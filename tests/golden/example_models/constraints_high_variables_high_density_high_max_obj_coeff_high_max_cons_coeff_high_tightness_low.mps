NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_274_0
 L  R_274_1
 L  R_274_2
 L  R_274_3
COLUMNS
    x_0       OBJROW     -8.           R_274_1   86.         
    x_0       R_274_3   28.         
    x_1       OBJROW     -12.          R_274_0   75.         
    x_1       R_274_1   56.            R_274_2   93.         
    x_2       OBJROW     -11.          R_274_0   48.         
    x_2       R_274_1   57.            R_274_2   5.          
    x_3       OBJROW     -47.          R_274_0   41.         
    x_3       R_274_2   68.            R_274_3   23.         
RHS
    RHS       R_274_0   175.           R_274_1   174.        
    RHS       R_274_2   161.           R_274_3   171.        
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
 UI BOUND     x_2       100.        
 UI BOUND     x_3       100.        
ENDATA
