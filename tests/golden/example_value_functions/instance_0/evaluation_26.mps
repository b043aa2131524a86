NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_158_0
 L  R_158_1
COLUMNS
    x_0       OBJROW     -4.        
    x_1       OBJROW     -6.           R_158_1   78.         
    x_2       OBJROW     -6.           R_158_0   5.          
    x_2       R_158_1   21.         
    x_3       OBJROW     -24.       
RHS
    RHS       R_158_0   103.           R_158_1   100.        
BOUNDS
 UI BOUND     x_0       43.         
 UI BOUND     x_1       43.         
 UI BOUND     x_2       43.         
 UI BOUND     x_3       43.         
ENDATA
