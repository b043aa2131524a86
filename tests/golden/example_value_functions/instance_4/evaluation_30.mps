NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_814_0
 L  R_814_1
COLUMNS
    x_0       OBJROW     -14.          R_814_0   11.         
    x_1       OBJROW     -14.          R_814_1   39.         
    x_2       OBJROW     -1.           R_814_0   3.          
    x_3       OBJROW     -2.           R_814_0   28.         
RHS
    RHS       R_814_0   4.             R_814_1   41.         
BOUNDS
 UI BOUND     x_0       14.         
 UI BOUND     x_1       14.         
 UI BOUND     x_2       14.         
 UI BOUND     x_3       14.         
ENDATA
