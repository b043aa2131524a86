NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_818_0
 L  R_818_1
COLUMNS
    x_0       OBJROW     -14.          R_818_0   11.         
    x_1       OBJROW     -14.          R_818_1   39.         
    x_2       OBJROW     -1.           R_818_0   3.          
    x_3       OBJROW     -2.           R_818_0   28.         
RHS
    RHS       R_818_0   25.            R_818_1   7.          
BOUNDS
 UI BOUND     x_0       14.         
 UI BOUND     x_1       14.         
 UI BOUND     x_2       14.         
 UI BOUND     x_3       14.         
ENDATA
