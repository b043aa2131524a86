NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_202_0
 L  R_202_1
COLUMNS
    x_0       OBJROW     -8.           R_202_1   5.          
    x_1       OBJROW     -12.          R_202_0   4.          
    x_1       R_202_1   10.         
    x_2       OBJROW     -11.          R_202_0   7.          
    x_2       R_202_1   9.          
    x_3       OBJROW     -47.          R_202_1   8.          
RHS
    RHS       R_202_0   20.            R_202_1   18.         
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
 UI BOUND     x_2       100.        
 UI BOUND     x_3       100.        
ENDATA
