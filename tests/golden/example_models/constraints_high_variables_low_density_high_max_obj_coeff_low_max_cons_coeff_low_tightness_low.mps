NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_186_0
 L  R_186_1
COLUMNS
    x_0       OBJROW     -1.           R_186_1   5.          
    x_1       OBJROW     -2.           R_186_0   4.          
    x_1       R_186_1   10.         
    x_2       OBJROW     -2.           R_186_0   7.          
    x_2       R_186_1   9.          
    x_3       OBJROW     -6.           R_186_1   8.          
RHS
    RHS       R_186_0   20.            R_186_1   18.         
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
 UI BOUND     x_2       10.         
 UI BOUND     x_3       10.         
ENDATA
