NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_98_0
 L  R_98_1
 L  R_98_2
 L  R_98_3
COLUMNS
    x_0       OBJROW     -1.           R_98_0    22.         
    x_1       OBJROW     -2.           R_98_3    56.         
RHS
    RHS       R_98_0    25.            R_98_1    23.         
    RHS       R_98_2    26.            R_98_3    25.         
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
