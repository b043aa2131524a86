NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_838_0
 L  R_838_1
COLUMNS
    x_0       OBJROW     -14.          R_838_0   11.         
    x_1       OBJROW     -14.          R_838_1   39.         
    x_2       OBJROW     -1.           R_838_0   3.          
    x_3       OBJROW     -2.           R_838_0   28.         
RHS
    RHS       R_838_0   11.            R_838_1   5.          
BOUNDS
 UI BOUND     x_0       14.         
 UI BOUND     x_1       14.         
 UI BOUND     x_2       14.         
 UI BOUND     x_3       14.         
ENDATA
