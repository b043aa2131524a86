NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_258_0
 L  R_258_1
 L  R_258_2
 L  R_258_3
COLUMNS
    x_0       OBJROW     -1.           R_258_1   86.         
    x_0       R_258_3   28.         
    x_1       OBJROW     -2.           R_258_0   75.         
    x_1       R_258_1   56.            R_258_2   93.         
    x_2       OBJROW     -2.           R_258_0   48.         
    x_2       R_258_1   57.            R_258_2   5.          
    x_3       OBJROW     -6.           R_258_0   41.         
    x_3       R_258_2   68.            R_258_3   23.         
RHS
    RHS       R_258_0   175.           R_258_1   174.        
    RHS       R_258_2   161.           R_258_3   171.        
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
 UI BOUND     x_2       10.         
 UI BOUND     x_3       10.         
ENDATA
