NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_74_0
 L  R_74_1
COLUMNS
    x_0       OBJROW     -8.           R_74_0    3.          
    x_0       R_74_1    5.          
    x_1       OBJROW     -12.          R_74_0    4.          
    x_1       R_74_1    10.         
RHS
    RHS       R_74_0    10.            R_74_1    8.          
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
ENDATA
