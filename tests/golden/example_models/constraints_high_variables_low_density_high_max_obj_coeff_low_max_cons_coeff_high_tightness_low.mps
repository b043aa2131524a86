NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_194_0
 L  R_194_1
COLUMNS
    x_0       OBJROW     -1.           R_194_1   86.         
    x_1       OBJROW     -2.           R_194_1   28.         
    x_2       OBJROW     -2.           R_194_0   75.         
    x_2       R_194_1   56.         
    x_3       OBJROW     -6.           R_194_0   93.         
RHS
    RHS       R_194_0   192.           R_194_1   183.        
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
 UI BOUND     x_2       10.         
 UI BOUND     x_3       10.         
ENDATA
