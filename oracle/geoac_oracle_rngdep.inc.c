/* TEST INFRASTRUCTURE ONLY (see geoac_oracle.h).  Included by geoac_oracle.c.
 *
 * Plain-C restatement of the range-dependent Cartesian atmosphere and equation set:
 *   Code/Atmo/G2S_MultiDimSpline3D.cpp      grid of .met profiles -> vertical natural splines of f, df/dx, df/dy per node,
 *                                           "bicubic of vertical splines" evaluation (Eval_Spline_f/df, AllOrder1/2)
 *   Code/GeoAc/GeoAc.EquationSets.3DRngDep.cpp   6/18-equation Cartesian moving-medium system
 *   Code/GeoAc3D.RngDep_main.cpp:244-328    fan / bounce / post-pass loops (in geoac_oracle.c: orc_fan)
 * Operand order, pow() calls, the full 16x16 matrix products and quirk Q11 (y-rows scaled by dx_scalar in the scalar
 * evaluators and in the d2f/dz2 block of AllOrder2) are kept so the compiled reference is reproduced bit for bit.
 */

/* G2S_MultiDimSpline3D.cpp:213-230 */
static const double BiCubic_ConversionMatrix[16][16] = {
    { 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    { 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {-3, 3, 0, 0,-2,-1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    { 2,-2, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    { 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0},
    { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0},
    { 0, 0, 0, 0, 0, 0, 0, 0,-3, 3, 0, 0,-2,-1, 0, 0},
    { 0, 0, 0, 0, 0, 0, 0, 0, 2,-2, 0, 0, 1, 1, 0, 0},
    {-3, 0, 3, 0, 0, 0, 0, 0,-2, 0,-1, 0, 0, 0, 0, 0},
    { 0, 0, 0, 0,-3, 0, 3, 0, 0, 0, 0, 0,-2, 0,-1, 0},
    { 9,-9,-9, 9, 6, 3,-6,-3, 6,-6, 3,-3, 4, 2, 2, 1},
    {-6, 6, 6,-6,-3,-3, 3, 3,-4, 4,-2, 2,-2,-2,-1,-1},
    { 2, 0,-2, 0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0},
    { 0, 0, 0, 0, 2, 0,-2, 0, 0, 0, 0, 0, 1, 0, 1, 0},
    {-6, 6, 6,-6,-4,-2, 4, 2,-3, 3,-3, 3,-2,-1,-2,-1},
    { 4,-4,-4, 4, 2, 2,-2,-2, 2,-2, 2,-2, 1, 1, 1, 1}
};

/* struct MultiDimSpline_3D: Code/GeoAc/G2S_MultiDimSpline3D.h:69-84 (flat arrays, index (ix*ny + iy)*nz + iz) */
typedef struct {
    int nx, ny, nz;
    int accel[3];
    const double *xv, *yv, *zv;
    double *f, *fs, *fxs, *fys;         /* values, f_slopes, dfdx_slopes, dfdy_slopes */
} mds3;

struct grid3d {
    int nx, ny, nz;
    double *xv, *yv, *zv;
    mds3 Temp, Windu, Windv, Dens;
    double x_min, x_max, y_min, y_max, z_min, z_max;
};

#define G3(S,ix,iy,iz) ((S)->f[((size_t)(ix) * (S)->ny + (iy)) * (S)->nz + (iz)])
#define G3S(S,A,ix,iy,iz) ((S)->A[((size_t)(ix) * (S)->ny + (iy)) * (S)->nz + (iz)])
#define IMIN(a,b) ((a) < (b) ? (a) : (b))
#define IMAX(a,b) ((a) > (b) ? (a) : (b))

/* one tridiagonal natural-spline solve along z: the three identical blocks of Set_Slopes_Multi (:313-343, :361-391, :394-424) */
static void slopes_along_z(const double* zv, int nz, const double* fcol, double* out, double* new_c, double* new_d){
    double ai, bi, ci, di;
    bi = 2.0 / (zv[1] - zv[0]);
    ci = 1.0 / (zv[1] - zv[0]);
    di = 3.0 * (fcol[1] - fcol[0]) / pow(zv[1] - zv[0], 2);
    new_c[0] = ci/bi;
    new_d[0] = di/bi;
    for(int i = 1; i < nz - 1; i++){
        ai = 1.0/(zv[i] - zv[i-1]);
        bi = 2.0 * (1.0/(zv[i] - zv[i-1]) + 1.0/(zv[i+1] - zv[i]));
        ci = 1.0/(zv[i+1] - zv[i]);
        di = 3.0 * ((fcol[i] - fcol[i-1]) / pow(zv[i] - zv[i-1], 2)
                    + (fcol[i+1] - fcol[i]) / pow(zv[i+1] - zv[i], 2) );
        new_c[i] = ci/(bi - new_c[i-1]*ai);
        new_d[i] = (di - new_d[i-1]*ai)/(bi - new_c[i-1]*ai);
    }
    ai = 1.0/(zv[nz-1] - zv[nz-2]);
    bi = 2.0/(zv[nz-1] - zv[nz-2]);
    di = 3.0 * (fcol[nz-1] - fcol[nz-2]) / pow(zv[nz-1] - zv[nz-2], 2);
    new_d[nz-1] = (di - new_d[nz - 2]*ai)/(bi - new_c[nz - 2]*ai);
    out[nz - 1] = new_d[nz - 1];
    for(int i = nz - 2; i >= 0; i--) out[i] = new_d[i] - new_c[i] * out[i+1];
}

/* Set_Slopes_Multi: G2S_MultiDimSpline3D.cpp:306-425 */
static void set_slopes_multi(mds3* S){
    int nx = S->nx, ny = S->ny, nz = S->nz;
    size_t ntot = (size_t)nx * ny * nz;
    double* new_c = (double*)malloc(sizeof(double) * (size_t)nz);
    double* new_d = (double*)malloc(sizeof(double) * (size_t)nz);
    double* dfdx = (double*)malloc(sizeof(double) * ntot);
    double* dfdy = (double*)malloc(sizeof(double) * ntot);
    for(int mx = 0; mx < nx; mx++) for(int my = 0; my < ny; my++)
        slopes_along_z(S->zv, nz, &G3(S, mx, my, 0), &G3S(S, fs, mx, my, 0), new_c, new_d);
    for(int mx = 0; mx < nx; mx++) for(int my = 0; my < ny; my++) for(int mz = 0; mz < nz; mz++){
        int mx_up = IMIN(mx + 1, nx - 1), mx_dn = IMAX(mx - 1, 0);
        int my_up = IMIN(my + 1, ny - 1), my_dn = IMAX(my - 1, 0);
        dfdx[((size_t)mx * ny + my) * nz + mz] = (G3(S, mx_up, my, mz) - G3(S, mx_dn, my, mz))/(S->xv[mx_up] - S->xv[mx_dn]);
        dfdy[((size_t)mx * ny + my) * nz + mz] = (G3(S, mx, my_up, mz) - G3(S, mx, my_dn, mz))/(S->yv[my_up] - S->yv[my_dn]);
    }
    for(int mx = 0; mx < nx; mx++) for(int my = 0; my < ny; my++)
        slopes_along_z(S->zv, nz, &dfdx[((size_t)mx * ny + my) * nz], &G3S(S, fxs, mx, my, 0), new_c, new_d);
    for(int mx = 0; mx < nx; mx++) for(int my = 0; my < ny; my++)
        slopes_along_z(S->zv, nz, &dfdy[((size_t)mx * ny + my) * nz], &G3S(S, fys, mx, my, 0), new_c, new_d);
    free(new_c); free(new_d); free(dfdx); free(dfdy);
}

/* ---- vertical spline evaluations at node (kx, ky), segment kz: G2S_MultiDimSpline3D.cpp:476-562 ---- */
static double ev_f(double z, const mds3* S, int kx, int ky, int kz){
    const double* zv = S->zv;
    double X = (z - zv[kz])/(zv[kz+1] - zv[kz]);
    double A = G3S(S, fs, kx, ky, kz) * (zv[kz+1] - zv[kz]) - (G3(S, kx, ky, kz+1) - G3(S, kx, ky, kz));
    double B = -G3S(S, fs, kx, ky, kz+1) * (zv[kz+1] - zv[kz]) + (G3(S, kx, ky, kz+1) - G3(S, kx, ky, kz));
    return (1.0 - X) * G3(S, kx, ky, kz) + X * G3(S, kx, ky, kz+1) + X * (1.0 - X) * (A * (1.0 - X ) + B * X);
}
static void fd_x_nodes(const mds3* S, int kx, int ky, int kz, double* a, double* b){
    int kx_up = IMIN(kx + 1, S->nx - 1), kx_dn = IMAX(kx - 1, 0);
    *a = (G3(S, kx_up, ky, kz) - G3(S, kx_dn, ky, kz))/(S->xv[kx_up] - S->xv[kx_dn]);
    *b = (G3(S, kx_up, ky, kz+1) - G3(S, kx_dn, ky, kz+1))/(S->xv[kx_up] - S->xv[kx_dn]);
}
static void fd_y_nodes(const mds3* S, int kx, int ky, int kz, double* a, double* b){
    int ky_up = IMIN(ky + 1, S->ny - 1), ky_dn = IMAX(ky - 1, 0);
    *a = (G3(S, kx, ky_up, kz) - G3(S, kx, ky_dn, kz))/(S->yv[ky_up] - S->yv[ky_dn]);
    *b = (G3(S, kx, ky_up, kz+1) - G3(S, kx, ky_dn, kz+1))/(S->yv[ky_up] - S->yv[ky_dn]);
}
static double ev_dfdx(double z, const mds3* S, int kx, int ky, int kz){
    const double* zv = S->zv;
    double dfdx_kz, dfdx_kzp1; fd_x_nodes(S, kx, ky, kz, &dfdx_kz, &dfdx_kzp1);
    double X = (z - zv[kz])/(zv[kz+1] - zv[kz]);
    double A = G3S(S, fxs, kx, ky, kz) * (zv[kz+1] - zv[kz]) - (dfdx_kzp1 - dfdx_kz);
    double B = -G3S(S, fxs, kx, ky, kz+1) * (zv[kz+1] - zv[kz]) + (dfdx_kzp1 - dfdx_kz);
    return (1.0 - X) * dfdx_kz + X * dfdx_kzp1 + X * (1.0 - X) * (A * (1.0 - X ) + B * X);
}
static double ev_dfdy(double z, const mds3* S, int kx, int ky, int kz){
    const double* zv = S->zv;
    double dfdy_kz, dfdy_kzp1; fd_y_nodes(S, kx, ky, kz, &dfdy_kz, &dfdy_kzp1);
    double X = (z - zv[kz])/(zv[kz+1] - zv[kz]);
    double A = G3S(S, fys, kx, ky, kz) * (zv[kz+1] - zv[kz]) - (dfdy_kzp1 - dfdy_kz);
    double B = -G3S(S, fys, kx, ky, kz+1) * (zv[kz+1] - zv[kz]) + (dfdy_kzp1 - dfdy_kz);
    return (1.0 - X) * dfdy_kz + X * dfdy_kzp1 + X * (1.0 - X) * (A * (1.0 - X ) + B * X);
}
static double ev_dfdz(double z, const mds3* S, int kx, int ky, int kz){
    const double* zv = S->zv;
    double X = (z - zv[kz])/(zv[kz+1] - zv[kz]);
    double A = G3S(S, fs, kx, ky, kz) * (zv[kz+1] - zv[kz]) - (G3(S, kx, ky, kz+1) - G3(S, kx, ky, kz));
    double B = -G3S(S, fs, kx, ky, kz+1) * (zv[kz+1] - zv[kz]) + (G3(S, kx, ky, kz+1) - G3(S, kx, ky, kz));
    return (G3(S, kx, ky, kz+1) - G3(S, kx, ky, kz))/(zv[kz+1] - zv[kz])
         + (1.0 - 2.0 * X) * (A * (1.0 - X) + B * X)/(zv[kz+1] - zv[kz])
         + X * (1.0 - X) * (B - A)/(zv[kz+1] - zv[kz]);
}
static double ev_ddfdxdz(double z, const mds3* S, int kx, int ky, int kz){
    const double* zv = S->zv;
    double dfdx_kz, dfdx_kzp1; fd_x_nodes(S, kx, ky, kz, &dfdx_kz, &dfdx_kzp1);
    double X = (z - zv[kz])/(zv[kz+1] - zv[kz]);
    double A = G3S(S, fxs, kx, ky, kz) * (zv[kz+1] - zv[kz]) - (dfdx_kzp1 - dfdx_kz);
    double B = -G3S(S, fxs, kx, ky, kz+1) * (zv[kz+1] - zv[kz]) + (dfdx_kzp1 - dfdx_kz);
    return (dfdx_kzp1 - dfdx_kz)/(zv[kz+1] - zv[kz])
         + (1.0 - 2.0 * X) * (A * (1.0 - X) + B * X)/(zv[kz+1] - zv[kz])
         + X * (1.0 - X) * (B - A)/(zv[kz+1] - zv[kz]);
}
static double ev_ddfdydz(double z, const mds3* S, int kx, int ky, int kz){
    const double* zv = S->zv;
    double dfdy_kz, dfdy_kzp1; fd_y_nodes(S, kx, ky, kz, &dfdy_kz, &dfdy_kzp1);
    double X = (z - zv[kz])/(zv[kz+1] - zv[kz]);
    double A = G3S(S, fys, kx, ky, kz) * (zv[kz+1] - zv[kz]) - (dfdy_kzp1 - dfdy_kz);
    double B = -G3S(S, fys, kx, ky, kz+1) * (zv[kz+1] - zv[kz]) + (dfdy_kzp1 - dfdy_kz);
    return (dfdy_kzp1 - dfdy_kz)/(zv[kz+1] - zv[kz])
         + (1.0 - 2.0 * X) * (A * (1.0 - X) + B * X)/(zv[kz+1] - zv[kz])
         + X * (1.0 - X) * (B - A)/(zv[kz+1] - zv[kz]);
}
static double ev_ddfdzdz(double z, const mds3* S, int kx, int ky, int kz){
    const double* zv = S->zv;
    double X = (z - zv[kz])/(zv[kz+1] - zv[kz]);
    double A = G3S(S, fs, kx, ky, kz) * (zv[kz+1] - zv[kz]) - (G3(S, kx, ky, kz+1) - G3(S, kx, ky, kz));
    double B = -G3S(S, fs, kx, ky, kz+1) * (zv[kz+1] - zv[kz]) + (G3(S, kx, ky, kz+1) - G3(S, kx, ky, kz));
    return 2.0 * (B - 2.0 * A + (A - B) * 3.0 * X)/pow(zv[kz+1] - zv[kz],2);
}

/* ---- horizontal finite differences of vertical-spline values ("BiCubic_Deriv_*"): :568-800 ---- */
typedef double (*evfn)(double, const mds3*, int, int, int);
static void updn(int k, int n, int* up, int* dn){ *up = k + 1; *dn = k - 1; if(*up > n - 1) *up = k; if(*dn < 0) *dn = k; }
static double fdx(evfn g, double z, const mds3* S, int kx, int ky, int kz){
    int up, dn; updn(kx, S->nx, &up, &dn);
    return (g(z, S, up, ky, kz) - g(z, S, dn, ky, kz))/(S->xv[up] - S->xv[dn]);
}
static double fdy(evfn g, double z, const mds3* S, int kx, int ky, int kz){
    int up, dn; updn(ky, S->ny, &up, &dn);
    return (g(z, S, kx, up, kz) - g(z, S, kx, dn, kz))/(S->yv[up] - S->yv[dn]);
}
static double fdxy(evfn g, double z, const mds3* S, int kx, int ky, int kz){
    int xu, xd, yu, yd; updn(kx, S->nx, &xu, &xd); updn(ky, S->ny, &yu, &yd);
    return (g(z, S, xu, yu, kz)
            - g(z, S, xu, yd, kz)
                - g(z, S, xd, yu, kz)
                    + g(z, S, xd, yd, kz))
                        /((S->xv[xu] - S->xv[xd])*(S->yv[yu] - S->yv[yd]));
}
#define BC_dfdx(z,S,i,j,k)         fdx(ev_f, z, S, i, j, k)
#define BC_dfdy(z,S,i,j,k)         fdy(ev_f, z, S, i, j, k)
#define BC_ddfdxdx(z,S,i,j,k)      fdx(ev_dfdx, z, S, i, j, k)
#define BC_ddfdydy(z,S,i,j,k)      fdy(ev_dfdy, z, S, i, j, k)
#define BC_ddfdxdy(z,S,i,j,k)      fdxy(ev_f, z, S, i, j, k)
#define BC_dddfdxdxdy(z,S,i,j,k)   fdxy(ev_dfdx, z, S, i, j, k)
#define BC_dddfdxdydy(z,S,i,j,k)   fdxy(ev_dfdy, z, S, i, j, k)
#define BC_dddfdxdydz(z,S,i,j,k)   fdxy(ev_dfdz, z, S, i, j, k)
#define BC_dddfdxdzdz(z,S,i,j,k)   fdx(ev_ddfdzdz, z, S, i, j, k)
#define BC_dddfdydzdz(z,S,i,j,k)   fdy(ev_ddfdzdz, z, S, i, j, k)
#define BC_ddddfdxdydzdz(z,S,i,j,k) fdxy(ev_ddfdzdz, z, S, i, j, k)

/* Find_Segment: G2S_MultiDimSpline3D.cpp:432-473 (same routine as the 1-D one) */
#define find_segment3 find_segment

static void bic_matvec(const double* X_vec, double* A_vec){
    for(int j = 0; j < 16; j++){
        A_vec[j] = 0;
        for(int k = 0; k < 16; k++) A_vec[j] += BiCubic_ConversionMatrix[j][k]*X_vec[k];
    }
}
static double bic_poly(const double* A_vec, double x_scaled, double y_scaled){
    double r = 0;
    for(int k1 = 0; k1 < 4; k1++){
    for(int k2 = 0; k2 < 4; k2++){
        r+=1.0*A_vec[k1 + 4*k2]*pow(x_scaled,k1)*pow(y_scaled,k2);
    }}
    return r;
}

/* Eval_Spline_f: :806-863 (inputs already clamped by the callers c(), u(), v(), rho()) */
static double g3_eval_f(double x, double y, double z, mds3* S){
    double A_vec[16], X_vec[16], BicCoeff[4][4];
    int kx = find_segment3(x, S->xv, S->nx, &S->accel[0]);
    int ky = find_segment3(y, S->yv, S->ny, &S->accel[1]);
    int kz = find_segment3(z, S->zv, S->nz, &S->accel[2]);
    double dx_scalar = S->xv[kx+1] - S->xv[kx];
    double dy_scalar = S->yv[ky+1] - S->yv[ky];
    static const int cx[4] = {0, 1, 0, 1}, cy[4] = {0, 0, 1, 1};
    for(int q = 0; q < 4; q++){
        X_vec[q]      = ev_f(z, S, kx + cx[q], ky + cy[q], kz);
        X_vec[4 + q]  = BC_dfdx(z, S, kx + cx[q], ky + cy[q], kz)*dx_scalar;
        X_vec[8 + q]  = BC_dfdy(z, S, kx + cx[q], ky + cy[q], kz)*dx_scalar;                 /* Q11: dx_scalar on the y rows (:829-832) */
        X_vec[12 + q] = BC_ddfdxdy(z, S, kx + cx[q], ky + cy[q], kz)*dx_scalar*dy_scalar;
    }
    bic_matvec(X_vec, A_vec);
    for(int k1 = 0; k1 < 4; k1++) for(int k2 = 0; k2 < 4; k2++) BicCoeff[k1][k2] = A_vec[k2*4 + k1];
    double x_scaled = (x - S->xv[kx])/(S->xv[kx+1] - S->xv[kx]);
    double y_scaled = (y - S->yv[ky])/(S->yv[ky+1] - S->yv[ky]);
    double result = 0;
    for(int k1 = 0; k1 < 4; k1++){
    for(int k2 = 0; k2 < 4; k2++){
        result+=1.0*BicCoeff[k1][k2]*pow(x_scaled,k1)*pow(y_scaled,k2);
    }}
    return result;
}

/* Eval_Spline_df: :865-966 */
static double g3_eval_df(double x, double y, double z, int index, mds3* S){
    double A_vec[16], X_vec[16], BicCoeff[4][4];
    int kx = find_segment3(x, S->xv, S->nx, &S->accel[0]);
    int ky = find_segment3(y, S->yv, S->ny, &S->accel[1]);
    int kz = find_segment3(z, S->zv, S->nz, &S->accel[2]);
    double dx_scalar = S->xv[kx+1] - S->xv[kx];
    double dy_scalar = S->yv[ky+1] - S->yv[ky];
    static const int cx[4] = {0, 1, 0, 1}, cy[4] = {0, 0, 1, 1};
    for(int q = 0; q < 4; q++){
        int i = kx + cx[q], j = ky + cy[q];
        if(index == 0){
            X_vec[q]      = BC_dfdx(z, S, i, j, kz);
            X_vec[4 + q]  = BC_ddfdxdx(z, S, i, j, kz)*dx_scalar;
            X_vec[8 + q]  = BC_ddfdxdy(z, S, i, j, kz)*dx_scalar;
            X_vec[12 + q] = BC_dddfdxdxdy(z, S, i, j, kz)*dx_scalar*dy_scalar;
        } else if(index == 1){
            X_vec[q]      = BC_dfdy(z, S, i, j, kz);
            X_vec[4 + q]  = BC_ddfdxdy(z, S, i, j, kz)*dx_scalar;
            X_vec[8 + q]  = BC_ddfdydy(z, S, i, j, kz)*dx_scalar;
            X_vec[12 + q] = BC_dddfdxdydy(z, S, i, j, kz)*dx_scalar*dy_scalar;
        } else {
            X_vec[q]      = ev_dfdz(z, S, i, j, kz);
            X_vec[4 + q]  = ev_ddfdxdz(z, S, i, j, kz)*dx_scalar;
            X_vec[8 + q]  = ev_ddfdydz(z, S, i, j, kz)*dx_scalar;
            X_vec[12 + q] = BC_dddfdxdydz(z, S, i, j, kz)*dx_scalar*dy_scalar;
        }
    }
    bic_matvec(X_vec, A_vec);
    for(int k1 = 0; k1 < 4; k1++) for(int k2 = 0; k2 < 4; k2++) BicCoeff[k1][k2] = A_vec[k2*4 + k1];
    double x_scaled = (x - S->xv[kx])/(S->xv[kx+1] - S->xv[kx]);
    double y_scaled = (y - S->yv[ky])/(S->yv[ky+1] - S->yv[ky]);
    double result = 0;
    for(int k1 = 0; k1 < 4; k1++){
    for(int k2 = 0; k2 < 4; k2++){
        result+=1.0*BicCoeff[k1][k2]*pow(x_scaled,k1)*pow(y_scaled,k2);
    }}
    return result;
}

/* Eval_Spline_AllOrder1 (order2 = 0, :1156-1339) and Eval_Spline_AllOrder2 (order2 = 1, :1341-1593).
 * out: f, dfdx, dfdy, dfdz [, ddfdxdx, ddfdydy, ddfdzdz, ddfdxdy, ddfdxdz, ddfdydz] */
static void g3_eval_all(const struct grid3d* G, double x, double y, double z, mds3* S, int order2, double* out){
    double A_vec[16], X_vec[16];
    double x_eval = DMIN(x, G->x_max);  x_eval = DMAX(x_eval, G->x_min);
    double y_eval = DMIN(y, G->y_max);  y_eval = DMAX(y_eval, G->y_min);
    double z_eval = DMIN(z, G->z_max);  z_eval = DMAX(z_eval, G->z_min);
    int kx = find_segment3(x_eval, S->xv, S->nx, &S->accel[0]);
    int ky = find_segment3(y_eval, S->yv, S->ny, &S->accel[1]);
    int kz = find_segment3(z_eval, S->zv, S->nz, &S->accel[2]);
    double dx_scalar = S->xv[kx+1] - S->xv[kx];
    double dy_scalar = S->yv[ky+1] - S->yv[ky];
    double x_scaled, y_scaled;
    if(order2){ x_scaled = (x_eval - S->xv[kx])/dx_scalar; y_scaled = (y_eval - S->yv[ky])/dy_scalar; }
    else { x_scaled = (x_eval - S->xv[kx])/(S->xv[kx+1] - S->xv[kx]); y_scaled = (y_eval - S->yv[ky])/(S->yv[ky+1] - S->yv[ky]); }
    static const int cx[4] = {0, 1, 0, 1}, cy[4] = {0, 0, 1, 1};
    double Fdx[4], Fdy[4], Fdxy[4];
    for(int q = 0; q < 4; q++) Fdx[q]  = BC_dfdx(z_eval, S, kx + cx[q], ky + cy[q], kz);
    for(int q = 0; q < 4; q++) Fdy[q]  = BC_dfdy(z_eval, S, kx + cx[q], ky + cy[q], kz);
    for(int q = 0; q < 4; q++) Fdxy[q] = BC_ddfdxdy(z_eval, S, kx + cx[q], ky + cy[q], kz);

    /* f */
    for(int q = 0; q < 4; q++){
        X_vec[q] = ev_f(z_eval, S, kx + cx[q], ky + cy[q], kz);
        X_vec[4 + q] = Fdx[q]*dx_scalar;
        X_vec[8 + q] = Fdy[q]*dy_scalar;
        X_vec[12 + q] = Fdxy[q]*dx_scalar*dy_scalar;
    }
    bic_matvec(X_vec, A_vec);
    out[0] = bic_poly(A_vec, x_scaled, y_scaled);

    /* df/dx (+ d2f/dx2, d2f/dxdy) */
    for(int q = 0; q < 4; q++){
        X_vec[q] = Fdx[q];
        X_vec[4 + q] = BC_ddfdxdx(z_eval, S, kx + cx[q], ky + cy[q], kz)*dx_scalar;
        X_vec[8 + q] = Fdxy[q]*dy_scalar;
        X_vec[12 + q] = BC_dddfdxdxdy(z_eval, S, kx + cx[q], ky + cy[q], kz)*dx_scalar*dy_scalar;
    }
    bic_matvec(X_vec, A_vec);
    out[1] = bic_poly(A_vec, x_scaled, y_scaled);
    if(order2){
        double ddfdxdx = 0;
        for(int k1 = 1; k1 < 4; k1++){
        for(int k2 = 0; k2 < 4; k2++){
            ddfdxdx+=1.0*k1*A_vec[k1 + 4*k2]*pow(x_scaled,k1-1)*pow(y_scaled,k2)/dx_scalar;
        }}
        double ddfdxdy = 0;
        for(int k1 = 0; k1 < 4; k1++){
        for(int k2 = 1; k2 < 4; k2++){
            ddfdxdy+=1.0*k2*A_vec[k1 + 4*k2]*pow(x_scaled,k1)*pow(y_scaled,k2-1)/dy_scalar;
        }}
        out[4] = ddfdxdx; out[7] = ddfdxdy;
    }

    /* df/dy (+ d2f/dy2) */
    for(int q = 0; q < 4; q++){
        X_vec[q] = Fdy[q];
        X_vec[4 + q] = Fdxy[q]*dx_scalar;
        X_vec[8 + q] = BC_ddfdydy(z_eval, S, kx + cx[q], ky + cy[q], kz)*dy_scalar;
        X_vec[12 + q] = BC_dddfdxdydy(z_eval, S, kx + cx[q], ky + cy[q], kz)*dx_scalar*dy_scalar;
    }
    bic_matvec(X_vec, A_vec);
    out[2] = bic_poly(A_vec, x_scaled, y_scaled);
    if(order2){
        double ddfdydy = 0;
        for(int k1 = 0; k1 < 4; k1++){
        for(int k2 = 1; k2 < 4; k2++){
            ddfdydy+=1.0*k2*A_vec[k1 + 4*k2]*pow(x_scaled,k1)*pow(y_scaled,k2-1)/dy_scalar;
        }}
        out[5] = ddfdydy;
    }

    /* df/dz (+ d2f/dxdz, d2f/dydz) */
    for(int q = 0; q < 4; q++){
        X_vec[q] = ev_dfdz(z_eval, S, kx + cx[q], ky + cy[q], kz);
        X_vec[4 + q] = ev_ddfdxdz(z_eval, S, kx + cx[q], ky + cy[q], kz)*dx_scalar;
        X_vec[8 + q] = ev_ddfdydz(z_eval, S, kx + cx[q], ky + cy[q], kz)*dy_scalar;
        X_vec[12 + q] = BC_dddfdxdydz(z_eval, S, kx + cx[q], ky + cy[q], kz)*dx_scalar*dy_scalar;
    }
    bic_matvec(X_vec, A_vec);
    out[3] = bic_poly(A_vec, x_scaled, y_scaled);
    if(order2){
        double ddfdxdz = 0;
        for(int k1 = 1; k1 < 4; k1++){
        for(int k2 = 0; k2 < 4; k2++){
            ddfdxdz+=1.0*k1*A_vec[k1 + 4*k2]*pow(x_scaled,k1-1)*pow(y_scaled,k2)/dx_scalar;
        }}
        double ddfdydz = 0;
        for(int k1 = 0; k1 < 4; k1++){
        for(int k2 = 1; k2 < 4; k2++){
            ddfdydz+=1.0*k2*A_vec[k1 + 4*k2]*pow(x_scaled,k1)*pow(y_scaled,k2-1)/dy_scalar;
        }}
        out[8] = ddfdxdz; out[9] = ddfdydz;

        /* d2f/dz2 (Q11: dx_scalar on the y rows, :1568-1571) */
        for(int q = 0; q < 4; q++){
            X_vec[q] = ev_ddfdzdz(z_eval, S, kx + cx[q], ky + cy[q], kz);
            X_vec[4 + q] = BC_dddfdxdzdz(z_eval, S, kx + cx[q], ky + cy[q], kz)*dx_scalar;
            X_vec[8 + q] = BC_dddfdydzdz(z_eval, S, kx + cx[q], ky + cy[q], kz)*dx_scalar;
            X_vec[12 + q] = BC_ddddfdxdydzdz(z_eval, S, kx + cx[q], ky + cy[q], kz)*dx_scalar*dy_scalar;
        }
        bic_matvec(X_vec, A_vec);
        out[6] = bic_poly(A_vec, x_scaled, y_scaled);
    }
}

/* ---- Atmo_State.h scalar API on the grid: G2S_MultiDimSpline3D.cpp:1633-1743 ---- */
static void g3_clamp(const struct grid3d* G, double* x, double* y, double* z){
    double e;
    e = DMIN(*x, G->x_max); *x = DMAX(e, G->x_min);
    e = DMIN(*y, G->y_max); *y = DMAX(e, G->y_min);
    e = DMIN(*z, G->z_max); *z = DMAX(e, G->z_min);
}
static double g3_rho(struct grid3d* G, double x, double y, double z){ g3_clamp(G, &x, &y, &z); return g3_eval_f(x, y, z, &G->Dens); }
static double g3_c(struct grid3d* G, double x, double y, double z){ g3_clamp(G, &x, &y, &z); return sqrt(gamR * g3_eval_f(x, y, z, &G->Temp)); }
static double g3_u(struct grid3d* G, double x, double y, double z){ g3_clamp(G, &x, &y, &z); return g3_eval_f(x, y, z, &G->Windu); }
static double g3_v(struct grid3d* G, double x, double y, double z){ g3_clamp(G, &x, &y, &z); return g3_eval_f(x, y, z, &G->Windv); }
static double g3_c_diff(struct grid3d* G, double x, double y, double z, int n){
    double xe = x, ye = y, ze = z; g3_clamp(G, &xe, &ye, &ze);
    return gamR / (2.0 * g3_c(G, x, y, z)) * g3_eval_df(xe, ye, ze, n, &G->Temp);
}
static double g3_u_diff(struct grid3d* G, double x, double y, double z, int n){ g3_clamp(G, &x, &y, &z); return g3_eval_df(x, y, z, n, &G->Windu); }
static double g3_v_diff(struct grid3d* G, double x, double y, double z, int n){ g3_clamp(G, &x, &y, &z); return g3_eval_df(x, y, z, n, &G->Windv); }

/* ---- loading: SetUp_G2S_Arrays + Load_G2S_Multi + Spline_Multi_G2S (:111-189, :1603-1621) ---- */
static void g3_free(struct grid3d* G){
    if(!G) return;
    mds3* S[4] = { &G->Temp, &G->Windu, &G->Windv, &G->Dens };
    for(int i = 0; i < 4; i++){ free(S[i]->f); free(S[i]->fs); free(S[i]->fxs); free(S[i]->fys); }
    free(G->xv); free(G->yv); free(G->zv); free(G);
}
static int count_newlines(const char* path){
    FILE* fp = fopen(path, "r"); if(!fp) return -1;
    int n = 0, ch; while((ch = fgetc(fp)) != EOF) if(ch == '\n') n++;
    fclose(fp); return n;
}
static struct grid3d* g3_alloc(int nx, int ny, int nz){
    struct grid3d* G = (struct grid3d*)calloc(1, sizeof(struct grid3d));
    G->nx = nx; G->ny = ny; G->nz = nz;
    G->xv = malloc(sizeof(double) * (size_t)nx); G->yv = malloc(sizeof(double) * (size_t)ny); G->zv = malloc(sizeof(double) * (size_t)nz);
    size_t ntot = (size_t)nx * ny * nz;
    mds3* S[4] = { &G->Temp, &G->Windu, &G->Windv, &G->Dens };
    for(int i = 0; i < 4; i++){
        S[i]->nx = nx; S[i]->ny = ny; S[i]->nz = nz; S[i]->accel[0] = S[i]->accel[1] = S[i]->accel[2] = 0;
        S[i]->xv = G->xv; S[i]->yv = G->yv; S[i]->zv = G->zv;
        S[i]->f = malloc(sizeof(double) * ntot); S[i]->fs = malloc(sizeof(double) * ntot);
        S[i]->fxs = malloc(sizeof(double) * ntot); S[i]->fys = malloc(sizeof(double) * ntot);
    }
    return G;
}
static void g3_finish(struct grid3d* G){
    G->x_min = G->xv[0]; G->x_max = G->xv[G->nx - 1];
    G->y_min = G->yv[0]; G->y_max = G->yv[G->ny - 1];
    G->z_min = G->zv[0]; G->z_max = G->zv[G->nz - 1];
    set_slopes_multi(&G->Temp); set_slopes_multi(&G->Windu);
    set_slopes_multi(&G->Dens); set_slopes_multi(&G->Windv);
}
/* z_grnd_at_load: the RngDep mains parse z_grnd= BEFORE loading, so it does enter the wind taper (width 0.05, :167-168) */
static struct grid3d* g3_load(const char* prefix, const char* locx, const char* locy, const char* format, double z_grnd_at_load){
    int nx = count_newlines(locx), ny = count_newlines(locy);
    char buf[512];
    snprintf(buf, sizeof buf, "%s%i.met", prefix, 0);
    int nz = count_newlines(buf);
    if(nx < 2 || ny < 2 || nz < 3) return NULL;
    int fmt;
    if(strncmp(format, "zTuvdp", 6) == 0) fmt = 0; else if(strncmp(format, "zuvwTdp", 7) == 0) fmt = 1; else return NULL;
    struct grid3d* G = g3_alloc(nx, ny, nz);
    FILE* fp = fopen(locx, "r"); for(int i = 0; i < nx; i++) if(fscanf(fp, "%lf", &G->xv[i]) != 1) G->xv[i] = 0; fclose(fp);
    fp = fopen(locy, "r"); for(int i = 0; i < ny; i++) if(fscanf(fp, "%lf", &G->yv[i]) != 1) G->yv[i] = 0; fclose(fp);
    for(int ix = 0; ix < nx; ix++) for(int iy = 0; iy < ny; iy++){
        snprintf(buf, sizeof buf, "%s%i.met", prefix, ix * ny + iy);
        fp = fopen(buf, "r");
        if(!fp){ g3_free(G); return NULL; }
        for(int iz = 0; iz < nz; iz++){
            double t[7] = {0,0,0,0,0,0,0}; int nt = fmt ? 7 : 6;
            for(int j = 0; j < nt; j++) if(fscanf(fp, "%lf", &t[j]) != 1) t[j] = 0.0;
            double zz, TT, uu, vv, rr;
            if(fmt == 0){ zz = t[0]; TT = t[1]; uu = t[2]; vv = t[3]; rr = t[4]; }
            else { zz = t[0]; uu = t[1]; vv = t[2]; TT = t[4]; rr = t[5]; }
            G->zv[iz] = zz;
            uu *= (2.0 / (1.0 + exp(-(G->zv[iz] - z_grnd_at_load)/0.05)) - 1.0) / 1000.0;
            vv *= (2.0 / (1.0 + exp(-(G->zv[iz] - z_grnd_at_load)/0.05)) - 1.0) / 1000.0;
            G3(&G->Temp, ix, iy, iz) = TT; G3(&G->Windu, ix, iy, iz) = uu; G3(&G->Windv, ix, iy, iz) = vv; G3(&G->Dens, ix, iy, iz) = rr;
        }
        fclose(fp);
    }
    g3_finish(G);
    return G;
}

/* ------------------------------------------------------------------------------------------ */
/* 3-D range-dependent Cartesian set: GeoAc.EquationSets.3DRngDep.cpp                           */
/* ------------------------------------------------------------------------------------------ */
/* GeoAc_SetInitialConditions: 3DRngDep.cpp:70-136 */
static void rd_set_ic(orc_ctx* c, double x0, double y0, double z0){
    src_rd* S = &c->RD; double* y = ROW(c, 0); struct grid3d* G = c->G3;
    S->src_loc[0] = x0; S->src_loc[1] = y0; S->src_loc[2] = z0;
    S->c0 = g3_c(G, x0, y0, z0);
    double MachComps[3] = { g3_u(G, x0, y0, z0)/S->c0, g3_v(G, x0, y0, z0)/S->c0, 0.0/S->c0 };
    double th = c->theta, ph = c->phi;
    double nu0[3]    = { cos(th)*cos(ph),  cos(th)*sin(ph), sin(th) };
    double mu0_th[3] = {-sin(th)*cos(ph), -sin(th)*sin(ph), cos(th) };
    double mu0_ph[3] = {-cos(th)*sin(ph),  cos(th)*cos(ph), 0.0 };
    double MachScalar = 1.0 + (nu0[0]*MachComps[0] + nu0[1]*MachComps[1] + nu0[2]*MachComps[2]);
    S->nu0 = 1.0/MachScalar;
    for(int i = 0; i < c->EqCnt; i++){
        if(i == 0) y[i] = x0;
        else if(i == 1) y[i] = y0;
        else if(i == 2) y[i] = z0;
        else if(i < 6) y[i] = nu0[i-3]/MachScalar;
        else if(i < 9 || (i >= 12 && i < 15)) y[i] = 0.0;
        else if(i < 12) y[i] = mu0_th[i-9]/MachScalar - nu0[i-9]/pow(MachScalar,2.0) * (mu0_th[0]*MachComps[0] + mu0_th[1]*MachComps[1] + mu0_th[2]*MachComps[2]);
        else y[i] = mu0_ph[i-15]/MachScalar - nu0[i-15]/pow(MachScalar,2.0) * (mu0_ph[0]*MachComps[0] + mu0_ph[1]*MachComps[1] + mu0_ph[2]*MachComps[2]);
    }
    for(int n = 0; n < 3; n++){ G->Temp.accel[n] = 0; G->Windu.accel[n] = 0; G->Windv.accel[n] = 0; }     /* :130-134 */
}

/* GeoAc_UpdateSources: 3DRngDep.cpp:218-326 */
static void rd_update_sources(orc_ctx* c, const double* cur){
    src_rd* S = &c->RD; struct grid3d* G = c->G3;
    double x = cur[0], y = cur[1], z = cur[2];
    double nu[3] = { cur[3], cur[4], cur[5] };
    double oT[10], oU[10], oV[10];
    if(!c->CalcAmp){
        g3_eval_all(G, x, y, z, &G->Temp, 0, oT);
        for(int n = 0; n < 3; n++){ G->Windu.accel[n] = G->Temp.accel[n]; G->Windv.accel[n] = G->Temp.accel[n]; }
        g3_eval_all(G, x, y, z, &G->Windu, 0, oU);
        g3_eval_all(G, x, y, z, &G->Windv, 0, oV);
        S->u = oU[0]; S->v = oV[0]; S->w = 0.0;
        for(int n = 0; n < 3; n++){ S->du[n] = oU[1+n]; S->dv[n] = oV[1+n]; }
        S->c = sqrt(gamR * oT[0]);
        for(int n = 0; n < 3; n++){ S->dc[n] = gamR / (2.0 * S->c) * oT[1+n]; S->dw[n] = 0.0; }
        S->nu_mag = sqrt(nu[0]*nu[0] + nu[1]*nu[1] + nu[2]*nu[2]);
        S->c_gr[0] = S->c*nu[0]/S->nu_mag + S->u;
        S->c_gr[1] = S->c*nu[1]/S->nu_mag + S->v;
        S->c_gr[2] = S->c*nu[2]/S->nu_mag + S->w;
        S->c_gr_mag = sqrt(pow(S->c_gr[0],2) + pow(S->c_gr[1],2) + pow(S->c_gr[2],2));
        return;
    }
    double Xl[2][3] = { { cur[6],  cur[7],  cur[8]  }, { cur[12], cur[13], cur[14] } };
    double ml[2][3] = { { cur[9],  cur[10], cur[11] }, { cur[15], cur[16], cur[17] } };
    double dtemp[3], ddtemp[3][3], ddWindu[3][3], ddWindv[3][3];
    g3_eval_all(G, x, y, z, &G->Temp, 1, oT);
    for(int n = 0; n < 3; n++){ G->Windu.accel[n] = G->Temp.accel[n]; G->Windv.accel[n] = G->Temp.accel[n]; }
    g3_eval_all(G, x, y, z, &G->Windu, 1, oU);
    g3_eval_all(G, x, y, z, &G->Windv, 1, oV);
    S->u = oU[0]; S->v = oV[0]; S->w = 0.0;
    for(int n = 0; n < 3; n++){ dtemp[n] = oT[1+n]; S->du[n] = oU[1+n]; S->dv[n] = oV[1+n]; }
    /* out: [4] xx, [5] yy, [6] zz, [7] xy, [8] xz, [9] yz */
    #define FILL_DD(dd, o) do { dd[0][0] = o[4]; dd[1][1] = o[5]; dd[2][2] = o[6]; dd[0][1] = o[7]; dd[0][2] = o[8]; dd[1][2] = o[9]; \
                                dd[1][0] = dd[0][1]; dd[2][0] = dd[0][2]; dd[2][1] = dd[1][2]; } while(0)
    FILL_DD(ddtemp, oT); FILL_DD(ddWindu, oU); FILL_DD(ddWindv, oV);
    #undef FILL_DD
    S->c = sqrt(gamR * oT[0]);
    for(int n = 0; n < 3; n++){
        S->dc[n] = gamR / (2.0 * S->c) * dtemp[n];
        S->dw[n] = 0.0;
        for(int a = 0; a < 2; a++){ S->ddc[n][a] = 0.0; S->ddu[n][a] = 0.0; S->ddv[n][a] = 0.0; S->ddw[n][a] = 0.0; }
        for(int m = 0; m < 3; m++){
            for(int a = 0; a < 2; a++){
                S->ddc[n][a] += Xl[a][m]*(gamR/(2.0*S->c) * ddtemp[n][m] - pow(gamR,2)/(4.0 * pow(S->c,3)) * dtemp[n]*dtemp[m]);
            }
            for(int a = 0; a < 2; a++) S->ddu[n][a] += Xl[a][m]*ddWindu[n][m];
            for(int a = 0; a < 2; a++) S->ddv[n][a] += Xl[a][m]*ddWindv[n][m];
            for(int a = 0; a < 2; a++) S->ddw[n][a] += Xl[a][m]*0.0;
        }
    }
    for(int a = 0; a < 2; a++){ S->dc[3+a] = 0.0; S->du[3+a] = 0.0; S->dv[3+a] = 0.0; S->dw[3+a] = 0.0; }
    for(int n = 0; n < 3; n++){
        for(int a = 0; a < 2; a++) S->dc[3+a] += Xl[a][n]*S->dc[n];
        for(int a = 0; a < 2; a++) S->du[3+a] += Xl[a][n]*S->du[n];
        for(int a = 0; a < 2; a++) S->dv[3+a] += Xl[a][n]*S->dv[n];
        for(int a = 0; a < 2; a++) S->dw[3+a] += Xl[a][n]*S->dw[n];
    }
    S->nu_mag = sqrt(nu[0]*nu[0] + nu[1]*nu[1] + nu[2]*nu[2]);
    for(int a = 0; a < 2; a++) S->dnu_mag[a] = (nu[0]*ml[a][0] + nu[1]*ml[a][1] + nu[2]*ml[a][2])/S->nu_mag;
    S->c_gr[0] = S->c*nu[0]/S->nu_mag + S->u;
    S->c_gr[1] = S->c*nu[1]/S->nu_mag + S->v;
    S->c_gr[2] = S->c*nu[2]/S->nu_mag + S->w;
    S->c_gr_mag = sqrt(pow(S->c_gr[0],2) + pow(S->c_gr[1],2) + pow(S->c_gr[2],2));
    for(int a = 0; a < 2; a++){
        double wind_d[3] = { S->du[3+a], S->dv[3+a], S->dw[3+a] };
        for(int i = 0; i < 3; i++)
            S->dc_gr[i][a] = nu[i]/S->nu_mag*S->dc[3+a] + S->c*ml[a][i]/S->nu_mag - S->c*nu[i]/pow(S->nu_mag,2) * S->dnu_mag[a] + wind_d[i];
        S->dc_gr_mag[a] = (S->c_gr[0]*S->dc_gr[0][a] + S->c_gr[1]*S->dc_gr[1][a] + S->c_gr[2]*S->dc_gr[2][a])/S->c_gr_mag;
    }
}

/* GeoAc_EvalSrcEq: 3DRngDep.cpp:331-393 */
static double rd_eval_src_eq(const orc_ctx* c, const double* y, int q){
    const src_rd* S = &c->RD;
    double nu[3] = { y[3], y[4], y[5] };
    if(q < 3) return S->c_gr[q]/S->c_gr_mag;
    if(q < 6){
        int i = q - 3;
        return -1.0/S->c_gr_mag*(S->nu_mag*S->dc[i] + nu[0]*S->du[i] + nu[1]*S->dv[i] + nu[2]*S->dw[i]);
    }
    int a = (q >= 12) ? 1 : 0;
    int i = (q - 6) % 3;
    if((q - 6) % 6 < 3)
        return S->dc_gr[i][a]/S->c_gr_mag - S->c_gr[i]/pow(S->c_gr_mag,2) * S->dc_gr_mag[a];
    const double* mu = a ? (y + 15) : (y + 9);
    return 1.0/pow(S->c_gr_mag,2) * S->dc_gr_mag[a]*(S->nu_mag*S->dc[i] + nu[0]*S->du[i] + nu[1]*S->dv[i] + nu[2]*S->dw[i])
         - 1.0/S->c_gr_mag*(S->dnu_mag[a]*S->dc[i] + S->nu_mag*S->ddc[i][a]
                            + mu[0]*S->du[i] + mu[1]*S->dv[i] + mu[2]*S->dw[i]
                            + nu[0]*S->ddu[i][a] + nu[1]*S->ddv[i][a] + nu[2]*S->ddw[i][a]);
}

/* GeoAc_BreakCheck / GroundCheck: 3DRngDep.cpp:451-472 */
static int rd_break_check(const orc_ctx* c, int k){
    const double* y = ROW(c, k);
    int check = 0;
    if(y[0] > c->x_max_limit) check = 1;
    if(y[0] < c->x_min_limit) check = 1;
    if(y[1] > c->y_max_limit) check = 1;
    if(y[1] < c->y_min_limit) check = 1;
    if(y[2] > c->vert_limit) check = 1;
    return check;
}
static int rd_ground_check(const orc_ctx* c, int k){ return ROW(c, k)[2] < c->z_grnd; }

/* travel-time / attenuation segment: 3DRngDep.cpp:478-542, 598-634 */
static double rd_tt_seg(orc_ctx* c, int n){
    struct grid3d* G = c->G3;
    const double* a = ROW(c, n); const double* b = ROW(c, n+1);
    double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
    double ds = sqrt(dx*dx + dy*dy + dz*dz);
    double x = a[0] + dx/2.0, y = a[1] + dy/2.0, z = a[2] + dz/2.0;
    double nu[3];
    nu[0] = a[3] + (b[3] - a[3])/2.0;
    nu[1] = a[4] + (b[4] - a[4])/2.0;
    nu[2] = a[5] + (b[5] - a[5])/2.0;
    double nu_mag = sqrt(nu[0]*nu[0] + nu[1]*nu[1] + nu[2]*nu[2]);
    double SndSpd = g3_c(G, x, y, z);
    double c_prop[3];
    c_prop[0] = SndSpd*nu[0]/nu_mag + g3_u(G, x, y, z);
    c_prop[1] = SndSpd*nu[1]/nu_mag + g3_v(G, x, y, z);
    c_prop[2] = SndSpd*nu[2]/nu_mag + 0.0;
    double c_prop_mag = sqrt(pow(c_prop[0],2) + pow(c_prop[1],2) + pow(c_prop[2],2));
    return ds/c_prop_mag;
}
static double rd_att_seg(orc_ctx* c, int n, double freq){
    struct grid3d* G = c->G3;
    const double* a = ROW(c, n); const double* b = ROW(c, n+1);
    double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
    double ds = sqrt(dx*dx + dy*dy + dz*dz);
    double x = a[0] + dx/2.0, y = a[1] + dy/2.0, z = a[2] + dz/2.0;
    /* SuthBass_Alpha(x, y, z, f) of Atmo_State.Absorption.cpp with the 3-D medium; reference state at (0, 0, z_grnd) */
    double c_g = g3_c(G, 0.0, 0.0, c->z_grnd), rho_g = g3_rho(G, 0.0, 0.0, c->z_grnd);
    double c_z = g3_c(G, x, y, z), rho_z = g3_rho(G, x, y, z);
    return suthbass_core(c, z, c_g, rho_g, c_z, rho_z, freq)*ds;
}

/* GeoAc_Jacobian / GeoAc_Amplitude: 3DRngDep.cpp:547-592 */
static double rd_jacobian(orc_ctx* c, int k){
    struct grid3d* G = c->G3; const double* y = ROW(c, k);
    double nu[3] = { y[3], y[4], y[5] };
    double nu_mag = sqrt(nu[0]*nu[0] + nu[1]*nu[1] + nu[2]*nu[2]);
    double SndSpd = g3_c(G, y[0], y[1], y[2]);
    double c_prop[3] = { SndSpd*nu[0]/nu_mag + g3_u(G, y[0], y[1], y[2]), SndSpd*nu[1]/nu_mag + g3_v(G, y[0], y[1], y[2]), SndSpd*nu[2]/nu_mag + 0.0 };
    double c_prop_mag = sqrt(pow(c_prop[0],2) + pow(c_prop[1],2) + pow(c_prop[2],2));
    double dxds = c_prop[0]/c_prop_mag, dyds = c_prop[1]/c_prop_mag, dzds = c_prop[2]/c_prop_mag;
    double dxdtheta = y[6], dydtheta = y[7], dzdtheta = y[8];
    double dxdphi = y[12], dydphi = y[13], dzdphi = y[14];
    return dxds*(dydtheta*dzdphi - dydphi*dzdtheta)
         - dxdtheta*(dyds*dzdphi - dzds*dydphi)
         + dxdphi*(dyds*dzdtheta - dzds*dydtheta);
}
static double rd_amplitude(orc_ctx* c, int k){
    const src_rd* S = &c->RD; struct grid3d* G = c->G3; const double* y = ROW(c, k);
    double x = y[0], yy = y[1], z = y[2];
    double x0 = S->src_loc[0], y0 = S->src_loc[1], z0 = S->src_loc[2];
    double nu[3] = { y[3], y[4], y[5] };
    double c0 = S->c0, SndSpd = g3_c(G, x, yy, z), Windu = g3_u(G, x, yy, z), Windv = g3_v(G, x, yy, z), Windw = 0.0;
    double Windu0 = g3_u(G, x0, y0, z0), Windv0 = g3_v(G, x0, y0, z0), Windw0 = 0.0;
    double nu_mag = (c0 - nu[0]*Windu - nu[1]*Windv - nu[2]*Windw)/SndSpd;
    double nu_mag0 = 1.0 - nu[0]*Windu0/c0 - nu[1]*Windv0/c0 - nu[2]*Windw0/c0;
    double c_prop[3]  = { SndSpd*nu[0]/nu_mag + Windu, SndSpd*nu[1]/nu_mag + Windv, SndSpd*nu[2]/nu_mag + Windw };
    double c_prop0[3] = { c0*cos(c->theta)*cos(c->phi) + Windu0, c0*cos(c->theta)*sin(c->phi) + Windv0, c0*sin(c->theta) + Windw0 };
    double c_prop_mag  = sqrt(pow(c_prop[0],2) + pow(c_prop[1],2) + pow(c_prop[2],2));
    double c_prop_mag0 = sqrt(pow(c_prop0[0],2) + pow(c_prop0[1],2) + pow(c_prop0[2],2));
    double D = rd_jacobian(c, k);
    double Amp_Num = g3_rho(G, x, yy, z) * nu_mag * pow(SndSpd,3) * c_prop_mag0 * cos(c->theta);
    double Amp_Den = g3_rho(G, x0, y0, z0) * nu_mag0 * pow(c0,3) * c_prop_mag * D;
    return 1.0/(4.0*Pi)*sqrt(fabs(Amp_Num/Amp_Den));
}

/* ApproximateIntercept + SetReflectionConditions: 3DRngDep.cpp:142-201 */
static void rd_reflect(orc_ctx* c, int k){
    const src_rd* S = &c->RD; struct grid3d* G = c->G3;
    double prev[18];
    const double* yk = ROW(c, k); const double* ykm = ROW(c, k-1); const double* ykmm = ROW(c, k-2);
    double zg = c->z_grnd;
    double dz_k = yk[2] - ykm[2];
    double dz_grnd = ykm[2] - zg;
    for(int i = 0; i < c->EqCnt; i++)
        prev[i] = ykm[i] + (ykm[i] - yk[i])/dz_k*dz_grnd
                + 1.0/2.0*(yk[i] + ykmm[i] - 2.0*ykm[i])/pow(dz_k,2.0)*pow(dz_grnd,2.0);
    double c_grnd = g3_c(G, prev[0], prev[1], zg);
    double dnuz_ds = - 1.0/c_grnd * (S->c0/c_grnd * g3_c_diff(G, prev[0], prev[1], zg, 2)
                                     + prev[3] * g3_u_diff(G, prev[0], prev[1], zg, 2)
                                     + prev[4] * g3_v_diff(G, prev[0], prev[1], zg, 2)
                                     + prev[5] * 0.0);
    double* y0 = ROW(c, 0);
    for(int i = 0; i < c->EqCnt; i++){
        if(i == 2) y0[i] = zg;
        else if(i == 5 || i == 8 || i == 14) y0[i] = -prev[i];
        else if(i == 11 || i == 17) y0[i] = -prev[i] + 2.0 * dnuz_ds*prev[i - 3]/(c_grnd/S->c0 * prev[5]);
        else y0[i] = prev[i];
    }
}
