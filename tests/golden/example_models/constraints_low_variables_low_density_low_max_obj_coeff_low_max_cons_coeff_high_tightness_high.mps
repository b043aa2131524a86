NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_38_0
 L  R_38_1
COLUMNS
    x_0       OBJROW     -1.           R_38_0    22.         
    x_1       OBJROW     -2.        
RHS
    RHS       R_38_0    24.            R_38_1    11.         
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
