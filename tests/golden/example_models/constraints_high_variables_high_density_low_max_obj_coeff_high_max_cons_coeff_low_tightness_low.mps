NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_234_0
 L  R_234_1
 L  R_234_2
 L  R_234_3
COLUMNS
    x_0       OBJROW     -8.        
    x_1       OBJROW     -12.          R_234_3   7.          
    x_2       OBJROW     -11.       
    x_3       OBJROW     -47.       
RHS
    RHS       R_234_0   4.             R_234_1   5.          
    RHS       R_234_2   5.             R_234_3   5.          
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
 UI BOUND     x_2       100.        
 UI BOUND     x_3       100.        
ENDATA
