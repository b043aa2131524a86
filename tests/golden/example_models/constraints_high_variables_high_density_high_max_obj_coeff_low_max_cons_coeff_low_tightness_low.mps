NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_250_0
 L  R_250_1
 L  R_250_2
 L  R_250_3
COLUMNS
    x_0       OBJROW     -1.           R_250_1   5.          
    x_0       R_250_2   4.             R_250_3   10.         
    x_1       OBJROW     -2.           R_250_0   7.          
    x_1       R_250_1   9.             R_250_3   8.          
    x_2       OBJROW     -2.           R_250_0   1.          
    x_2       R_250_2   6.          
    x_3       OBJROW     -6.           R_250_0   9.          
    x_3       R_250_1   3.             R_250_2   1.          
    x_3       R_250_3   3.          
RHS
    RHS       R_250_0   17.            R_250_1   20.         
    RHS       R_250_2   20.            R_250_3   18.         
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
 UI BOUND     x_2       10.         
 UI BOUND     x_3       10.         
ENDATA
