NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_246_0
 L  R_246_1
 L  R_246_2
 L  R_246_3
COLUMNS
    x_0       OBJROW     -8.        
    x_1       OBJROW     -12.          R_246_3   56.         
    x_2       OBJROW     -11.       
    x_3       OBJROW     -47.       
RHS
    RHS       R_246_0   11.            R_246_1   33.         
    RHS       R_246_2   39.            R_246_3   30.         
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
 UI BOUND     x_2       100.        
 UI BOUND     x_3       100.        
ENDATA
