NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_386_0
 L  R_386_1
COLUMNS
    x_0       OBJROW     -8.        
    x_1       OBJROW     -12.          R_386_1   7.          
    x_2       OBJROW     -11.       
    x_3       OBJROW     -47.          R_386_0   14.         
RHS
    RHS       R_386_0   8.             R_386_1   16.         
BOUNDS
 UI BOUND     x_0       87.         
 UI BOUND     x_1       87.         
 UI BOUND     x_2       87.         
 UI BOUND     x_3       87.         
ENDATA
