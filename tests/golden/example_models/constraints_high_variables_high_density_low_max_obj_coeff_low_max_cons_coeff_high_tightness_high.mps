NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_230_0
 L  R_230_1
 L  R_230_2
 L  R_230_3
COLUMNS
    x_0       OBJROW     -1.        
    x_1       OBJROW     -2.           R_230_3   56.         
    x_2       OBJROW     -2.        
    x_3       OBJROW     -6.        
RHS
    RHS       R_230_0   11.            R_230_1   33.         
    RHS       R_230_2   39.            R_230_3   30.         
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
 UI BOUND     x_2       10.         
 UI BOUND     x_3       10.         
ENDATA
