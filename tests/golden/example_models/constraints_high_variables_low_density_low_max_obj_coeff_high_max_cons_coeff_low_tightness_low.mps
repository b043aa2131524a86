NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_170_0
 L  R_170_1
COLUMNS
    x_0       OBJROW     -8.        
    x_1       OBJROW     -12.       
    x_2       OBJROW     -11.       
    x_3       OBJROW     -47.          R_170_1   7.          
RHS
    RHS       R_170_0   5.             R_170_1   5.          
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
 UI BOUND     x_2       100.        
 UI BOUND     x_3       100.        
ENDATA
