NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_262_0
 L  R_262_1
 L  R_262_2
 L  R_262_3
COLUMNS
    x_0       OBJROW     -1.           R_262_1   86.         
    x_0       R_262_3   28.         
    x_1       OBJROW     -2.           R_262_0   75.         
    x_1       R_262_1   56.            R_262_2   93.         
    x_2       OBJROW     -2.           R_262_0   48.         
    x_2       R_262_1   57.            R_262_2   5.          
    x_3       OBJROW     -6.           R_262_0   41.         
    x_3       R_262_2   68.            R_262_3   23.         
RHS
    RHS       R_262_0   100.           R_262_1   99.         
    RHS       R_262_2   46.            R_262_3   85.         
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
 UI BOUND     x_2       10.         
 UI BOUND     x_3       10.         
ENDATA
