NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_150_0
 L  R_150_1
 L  R_150_2
 L  R_150_3
COLUMNS
    x_0       OBJROW     -8.           R_150_0   22.         
    x_0       R_150_1   86.            R_150_3   28.         
    x_1       OBJROW     -12.          R_150_0   75.         
    x_1       R_150_1   56.            R_150_2   93.         
RHS
    RHS       R_150_0   85.            R_150_1   67.         
    RHS       R_150_2   89.            R_150_3   76.         
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
ENDATA
