NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_226_0
 L  R_226_1
 L  R_226_2
 L  R_226_3
COLUMNS
    x_0       OBJROW     -1.        
    x_1       OBJROW     -2.           R_226_3   56.         
    x_2       OBJROW     -2.        
    x_3       OBJROW     -6.        
RHS
    RHS       R_226_0   53.            R_226_1   40.         
    RHS       R_226_2   45.            R_226_3   47.         
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
 UI BOUND     x_2       10.         
 UI BOUND     x_3       10.         
ENDATA
