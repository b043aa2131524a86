NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_230_0
 L  R_230_1
COLUMNS
    x_0       OBJROW     -2.           R_230_1   20.         
    x_1       OBJROW     -3.           R_230_0   14.         
    x_1       R_230_1   38.         
    x_2       OBJROW     -3.           R_230_0   28.         
    x_2       R_230_1   33.         
    x_3       OBJROW     -12.          R_230_1   29.         
RHS
    RHS       R_230_0   45.            R_230_1   81.         
BOUNDS
 UI BOUND     x_0       26.         
 UI BOUND     x_1       26.         
 UI BOUND     x_2       26.         
 UI BOUND     x_3       26.         
ENDATA
