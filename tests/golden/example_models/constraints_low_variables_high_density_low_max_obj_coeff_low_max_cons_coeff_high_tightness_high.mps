NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_102_0
 L  R_102_1
 L  R_102_2
 L  R_102_3
COLUMNS
    x_0       OBJROW     -1.           R_102_0   22.         
    x_1       OBJROW     -2.           R_102_3   56.         
RHS
    RHS       R_102_0   25.            R_102_1   17.         
    RHS       R_102_2   21.            R_102_3   16.         
BOUNDS
 UI BOUND     x_0       10.         
 UI BOUND     x_1       10.         
ENDATA
