NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_266_0
 L  R_266_1
 L  R_266_2
 L  R_266_3
COLUMNS
    x_0       OBJROW     -8.           R_266_1   5.          
    x_0       R_266_2   4.             R_266_3   10.         
    x_1       OBJROW     -12.          R_266_0   7.          
    x_1       R_266_1   9.             R_266_3   8.          
    x_2       OBJROW     -11.          R_266_0   1.          
    x_2       R_266_2   6.          
    x_3       OBJROW     -47.          R_266_0   9.          
    x_3       R_266_1   3.             R_266_2   1.          
    x_3       R_266_3   3.          
RHS
    RHS       R_266_0   17.            R_266_1   20.         
    RHS       R_266_2   20.            R_266_3   18.         
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
 UI BOUND     x_2       100.        
 UI BOUND     x_3       100.        
ENDATA
