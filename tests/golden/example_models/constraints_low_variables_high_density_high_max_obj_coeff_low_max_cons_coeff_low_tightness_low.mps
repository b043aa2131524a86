NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_122_0
 L  R_122_1
 L  R_122_2
 L  R_122_3
COLUMNS
    x_0       OBJROW     -1.           R_122_0   3.          
    x_0       R_122_1   5.             R_122_2   4.          
    x_0       R_122_3   10.         
    x_1       OBJROW     -2.           R_122_0   7.          
    x_1       R_122_1   9.             R_122_3   8.          
RHS
    RHS       R_122_0   10.            R_122_1   9.          
    RHS       R_122_2   8.             R_122_3   8.          
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
