NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_154_0
 L  R_154_1
COLUMNS
    x_0       OBJROW     -1.        
    x_1       OBJROW     -2.        
    x_2       OBJROW     -2.        
    x_3       OBJROW     -6.           R_154_1   7.          
RHS
    RHS       R_154_0   5.             R_154_1   5.          
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
 UI BOUND     x_2       10.         
 UI BOUND     x_3       10.         
ENDATA
