NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_602_0
 L  R_602_1
COLUMNS
    x_0       OBJROW     -2.        
    x_1       OBJROW     -3.           R_602_1   78.         
    x_2       OBJROW     -3.           R_602_0   5.          
    x_2       R_602_1   21.         
    x_3       OBJROW     -12.       
RHS
    RHS       R_602_0   128.           R_602_1   43.         
BOUNDS
 UI BOUND     x_0       23.         
 UI BOUND     x_1       23.         
 UI BOUND     x_2       23.         
 UI BOUND     x_3       23.         
ENDATA
