NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_194_0
 L  R_194_1
COLUMNS
    x_0       OBJROW     -4.        
    x_1       OBJROW     -6.           R_194_1   78.         
    x_2       OBJROW     -6.           R_194_0   5.          
    x_2       R_194_1   21.         
    x_3       OBJROW     -24.       
RHS
    RHS       R_194_0   124.           R_194_1   107.        
BOUNDS
 UI BOUND     x_0       43.         
 UI BOUND     x_1       43.         
 UI BOUND     x_2       43.         
 UI BOUND     x_3       43.         
ENDATA
