NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_114_0
 L  R_114_1
 L  R_114_2
 L  R_114_3
COLUMNS
    x_0       OBJROW     -8.           R_114_0   22.         
    x_1       OBJROW     -12.          R_114_3   56.         
RHS
    RHS       R_114_0   25.            R_114_1   23.         
    RHS       R_114_2   26.            R_114_3   25.         
BOUNDS
 UI BOUND     x_0       100.        
 UI BOUND     x_1       100.        
ENDATA
