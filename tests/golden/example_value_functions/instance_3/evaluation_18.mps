NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_606_0
 L  R_606_1
COLUMNS
    x_0       OBJROW     -2.        
    x_1       OBJROW     -3.           R_606_1   78.         
    x_2       OBJROW     -3.           R_606_0   5.          
    x_2       R_606_1   21.         
    x_3       OBJROW     -12.       
RHS
    RHS       R_606_0   110.           R_606_1   48.         
BOUNDS
 UI BOUND     x_0       23.         
 UI BOUND     x_1       23.         
 UI BOUND     x_2       23.         
 UI BOUND     x_3       23.         
ENDATA
