NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_218_0
 L  R_218_1
 L  R_218_2
 L  R_218_3
COLUMNS
    x_0       OBJROW     -1.        
    x_1       OBJROW     -2.           R_218_3   7.          
    x_2       OBJROW     -2.        
    x_3       OBJROW     -6.        
RHS
    RHS       R_218_0   4.             R_218_1   5.          
    RHS       R_218_2   5.             R_218_3   5.          
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
 UI BOUND     x_2       10.         
 UI BOUND     x_3       10.         
ENDATA
