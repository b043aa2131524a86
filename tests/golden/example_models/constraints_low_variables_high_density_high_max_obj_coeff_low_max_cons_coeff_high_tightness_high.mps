NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_134_0
 L  R_134_1
 L  R_134_2
 L  R_134_3
COLUMNS
    x_0       OBJROW     -1.           R_134_0   22.         
    x_0       R_134_1   86.            R_134_3   28.         
    x_1       OBJROW     -2.           R_134_0   75.         
    x_1       R_134_1   56.            R_134_2   93.         
RHS
    RHS       R_134_0   85.            R_134_1   67.         
    RHS       R_134_2   89.            R_134_3   76.         
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
