NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_238_0
 L  R_238_1
 L  R_238_2
 L  R_238_3
COLUMNS
    x_0       OBJROW     -8.        
    x_1       OBJROW     -12.          R_238_3   7.          
    x_2       OBJROW     -11.       
    x_3       OBJROW     -47.       
RHS
    RHS       R_238_0   1.             R_238_1   3.          
    RHS       R_238_2   4.             R_238_3   3.          
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
 UI BOUND     x_2       100.        
 UI BOUND     x_3       100.        
ENDATA
