NAME          ClpDefau
ROWS
 N  OBJROW
 L  R_66_0
 L  R_66_1
COLUMNS
    x_0       OBJROW     -4.        
    x_1       OBJROW     -6.           R_66_1    78.         
    x_2       OBJROW     -6.           R_66_0    5.          
    x_2       R_66_1    21.         
    x_3       OBJROW     -24.       
RHS
    RHS       R_66_0    96.            R_66_1    15.         
BOUNDS
 UI BOUND     x_0       43.         
 UI BOUND     x_1       43.         
 UI BOUND     x_2       43.         
 UI BOUND     x_3       43.         
ENDATA
