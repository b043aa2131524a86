NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_586_0
 L  R_586_1
COLUMNS
    x_0       OBJROW     -2.        
    x_1       OBJROW     -3.           R_586_1   78.         
    x_2       OBJROW     -3.           R_586_0   5.          
    x_2       R_586_1   21.         
    x_3       OBJROW     -12.       
RHS
    RHS       R_586_0   55.            R_586_1   109.        
BOUNDS
 UI BOUND     x_0       23.         
 UI BOUND     x_1       23.         
 UI BOUND     x_2       23.         
 UI BOUND     x_3       23.         
ENDATA
