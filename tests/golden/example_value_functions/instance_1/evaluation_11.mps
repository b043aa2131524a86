NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_258_0
 L  R_258_1
COLUMNS
    x_0       OBJROW     -2.           R_258_1   20.         
    x_1       OBJROW     -3.           R_258_0   14.         
    x_1       R_258_1   38.         
    x_2       OBJROW     -3.           R_258_0   28.         
    x_2       R_258_1   33.         
    x_3       OBJROW     -12.          R_258_1   29.         
RHS
    RHS       R_258_0   10.            R_258_1   11.         
BOUNDS
 UI BOUND     x_0       26.         
 UI BOUND     x_1       26.         
 UI BOUND     x_2       26.         
 UI BOUND     x_3       26.         
ENDATA
