NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_206_0
 L  R_206_1
COLUMNS
    x_0       OBJROW     -8.           R_206_1   5.          
    x_1       OBJROW     -12.          R_206_0   4.          
    x_1       R_206_1   10.         
    x_2       OBJROW     -11.          R_206_0   7.          
    x_2       R_206_1   9.          
    x_3       OBJROW     -47.          R_206_1   8.          
RHS
    RHS       R_206_0   20.            R_206_1   12.         
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
 UI BOUND     x_2       100.        
 UI BOUND     x_3       100.        
ENDATA
