NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_270_0
 L  R_270_1
 L  R_270_2
 L  R_270_3
COLUMNS
    x_0       OBJROW     -8.           R_270_1   5.          
    x_0       R_270_2   4.             R_270_3   10.         
    x_1       OBJROW     -12.          R_270_0   7.          
    x_1       R_270_1   9.             R_270_3   8.          
    x_2       OBJROW     -11.          R_270_0   1.          
    x_2       R_270_2   6.          
    x_3       OBJROW     -47.          R_270_0   9.          
    x_3       R_270_1   3.             R_270_2   1.          
    x_3       R_270_3   3.          
RHS
    RHS       R_270_0   8.             R_270_1   20.         
    RHS       R_270_2   20.            R_270_3   15.         
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
 UI BOUND     x_2       100.        
 UI BOUND     x_3       100.        
ENDATA
