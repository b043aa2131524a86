NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_222_0
 L  R_222_1
 L  R_222_2
 L  R_222_3
COLUMNS
    x_0       OBJROW     -1.        
    x_1       OBJROW     -2.           R_222_3   7.          
    x_2       OBJROW     -2.        
    x_3       OBJROW     -6.        
RHS
    RHS       R_222_0   1.             R_222_1   3.          
    RHS       R_222_2   4.             R_222_3   3.          
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
 UI BOUND     x_2       10.         
 UI BOUND     x_3       10.         
ENDATA
