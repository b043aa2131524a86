NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_198_0
 L  R_198_1
COLUMNS
    x_0       OBJROW     -1.           R_198_1   86.         
    x_1       OBJROW     -2.           R_198_1   28.         
    x_2       OBJROW     -2.           R_198_0   75.         
    x_2       R_198_1   56.         
    x_3       OBJROW     -6.           R_198_0   93.         
RHS
    RHS       R_198_0   170.           R_198_1   135.        
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
 UI BOUND     x_2       10.         
 UI BOUND     x_3       10.         
ENDATA
