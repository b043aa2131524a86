NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_214_0
 L  R_214_1
COLUMNS
    x_0       OBJROW     -8.           R_214_1   86.         
    x_1       OBJROW     -12.          R_214_1   28.         
    x_2       OBJROW     -11.          R_214_0   75.         
    x_2       R_214_1   56.         
    x_3       OBJROW     -47.          R_214_0   93.         
RHS
    RHS       R_214_0   170.           R_214_1   135.        
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
 UI BOUND     x_2       100.        
 UI BOUND     x_3       100.        
ENDATA
