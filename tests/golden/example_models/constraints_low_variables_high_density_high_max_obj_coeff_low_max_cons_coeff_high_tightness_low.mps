NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_130_0
 L  R_130_1
 L  R_130_2
 L  R_130_3
COLUMNS
    x_0       OBJROW     -1.           R_130_0   22.         
    x_0       R_130_1   86.            R_130_3   28.         
    x_1       OBJROW     -2.           R_130_0   75.         
    x_1       R_130_1   56.            R_130_2   93.         
RHS
    RHS       R_130_0   96.            R_130_1   91.         
    RHS       R_130_2   97.            R_130_3   94.         
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
