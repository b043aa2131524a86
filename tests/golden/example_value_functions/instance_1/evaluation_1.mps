NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_218_0
 L  R_218_1
COLUMNS
    x_0       OBJROW     -2.           R_218_1   20.         
    x_1       OBJROW     -3.           R_218_0   14.         
    x_1       R_218_1   38.         
    x_2       OBJROW     -3.           R_218_0   28.         
    x_2       R_218_1   33.         
    x_3       OBJROW     -12.          R_218_1   29.         
RHS
    RHS       R_218_0   13.            R_218_1   68.         
BOUNDS
 UI BOUND     x_0       26.         
 UI BOUND     x_1       26.         
 UI BOUND     x_2       26.         
 UI BOUND     x_3       26.         
ENDATA
