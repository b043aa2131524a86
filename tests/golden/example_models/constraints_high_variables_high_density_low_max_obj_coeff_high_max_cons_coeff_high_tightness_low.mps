NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_242_0
 L  R_242_1
 L  R_242_2
 L  R_242_3
COLUMNS
    x_0       OBJROW     -8.        
    x_1       OBJROW     -12.          R_242_3   56.         
    x_2       OBJROW     -11.       
    x_3       OBJROW     -47.       
RHS
    RHS       R_242_0   53.            R_242_1   40.         
    RHS       R_242_2   45.            R_242_3   47.         
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
 UI BOUND     x_2       100.        
 UI BOUND     x_3       100.        
ENDATA
