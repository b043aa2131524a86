NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_126_0
 L  R_126_1
 L  R_126_2
 L  R_126_3
COLUMNS
    x_0       OBJROW     -1.           R_126_0   3.          
    x_0       R_126_1   5.             R_126_2   4.          
    x_0       R_126_3   10.         
    x_1       OBJROW     -2.           R_126_0   7.          
    x_1       R_126_1   9.             R_126_3   8.          
RHS
    RHS       R_126_0   10.            R_126_1   6.          
    RHS       R_126_2   2.             R_126_3   2.          
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
