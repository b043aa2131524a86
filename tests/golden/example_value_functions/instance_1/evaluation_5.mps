NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_234_0
 L  R_234_1
COLUMNS
    x_0       OBJROW     -2.           R_234_1   20.         
    x_1       OBJROW     -3.           R_234_0   14.         
    x_1       R_234_1   38.         
    x_2       OBJROW     -3.           R_234_0   28.         
    x_2       R_234_1   33.         
    x_3       OBJROW     -12.          R_234_1   29.         
RHS
    RHS       R_234_0   65.            R_234_1   25.         
BOUNDS
 UI BOUND     x_0       26.         
 UI BOUND     x_1       26.         
 UI BOUND     x_2       26.         
 UI BOUND     x_3       26.         
ENDATA
