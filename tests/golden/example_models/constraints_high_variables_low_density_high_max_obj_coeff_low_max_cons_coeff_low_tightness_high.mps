NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_190_0
 L  R_190_1
COLUMNS
    x_0       OBJROW     -1.           R_190_1   5.          
    x_1       OBJROW     -2.           R_190_0   4.          
    x_1       R_190_1   10.         
    x_2       OBJROW     -2.           R_190_0   7.          
    x_2       R_190_1   9.          
    x_3       OBJROW     -6.           R_190_1   8.          
RHS
    RHS       R_190_0   20.            R_190_1   12.         
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
 UI BOUND     x_2       10.         
 UI BOUND     x_3       10.         
ENDATA
