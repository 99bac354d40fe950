/* TEST INFRASTRUCTURE ONLY (see geoac_oracle.h).  Included by geoac_oracle.c (after geoac_oracle_rngdep.inc.c).
 *
 * Plain-C restatement of the range-dependent spherical atmosphere and equation set:
 *   Code/Atmo/G2S_GlobalMultiDimSpline3D.cpp     grid of .met profiles on (lat, lon) -> vertical natural splines of f, df/dt, df/dp
 *                                                per node, "bicubic of vertical splines" evaluation (Eval_Spline_f/df, AllOrder1/2)
 *   Code/GeoAc/GeoAc.EquationSets.GlobalRngDep.cpp   6/18-equation spherical moving-medium system
 *   Code/GeoAcGlobal.RngDep_main.cpp:251-337     fan / bounce / post-pass loops (in geoac_oracle.c: orc_fan)
 * Arrays are [r][t][p] (t = latitude, p = longitude, radians).  Operand order, pow() calls and the full 16x16 matrix
 * products are kept, and so are the quirks (Q12):
 *   a. interior rows of the df/dt and df/dp slope systems use (d[i] - d[i+1]) where (d[i] - d[i-1]) is meant (:381, :414);
 *   b. Eval_Vert_Spline_ddfdrdt / ddfdrdp lead with (x - x) = 0 (:546, :562);
 *   c. AllOrder2's mixed second derivatives are not divided by the cell sizes (:1328-1338, :1374-1384, :1420-1424).
 */

typedef struct {
    int nr, nt, np;
    int accel[3];
    const double *rv, *tv, *pv;
    double *f, *fs, *fts, *fps;         /* values, f_slopes, dfdt_slopes, dfdp_slopes */
} mdsg;

struct gridg {
    int nr, nt, np;
    double *rv, *tv, *pv;
    mdsg Temp, Windu, Windv, Dens;
    double r_min, r_max, t_min, t_max, p_min, p_max;
};

#define GG(S,ir,it,ip) ((S)->f[((size_t)(ir) * (S)->nt + (it)) * (S)->np + (ip)])
#define GGS(S,A,ir,it,ip) ((S)->A[((size_t)(ir) * (S)->nt + (it)) * (S)->np + (ip)])

/* one tridiagonal solve along r for the column at (mt, mp): the three blocks of Set_Slopes_Multi (:320-349, :365-395, :398-428);
 * quirk = 1 reproduces the interior right-hand side of the df/dt and df/dp blocks (Q12a) */
static void slopes_along_r(const double* rv, int nr, const double* f0, size_t stride, double* out0, double* new_c, double* new_d, int quirk){
    #define FC(i) f0[(size_t)(i) * stride]
    double ai, bi, ci, di;
    bi = 2.0 / (rv[1] - rv[0]);
    ci = 1.0 / (rv[1] - rv[0]);
    di = 3.0 * (FC(1) - FC(0)) / pow(rv[1] - rv[0], 2);
    new_c[0] = ci/bi;
    new_d[0] = di/bi;
    for(int i = 1; i < nr - 1; i++){
        ai = 1.0/(rv[i] - rv[i-1]);
        bi = 2.0 * (1.0/(rv[i] - rv[i-1]) + 1.0/(rv[i+1] - rv[i]));
        ci = 1.0/(rv[i+1] - rv[i]);
        if(quirk)
            di = 3.0 * ((FC(i) - FC(i+1)) / pow(rv[i] - rv[i-1], 2)
                        + (FC(i+1) - FC(i)) / pow(rv[i+1] - rv[i], 2) );
        else
            di = 3.0 * ((FC(i) - FC(i-1)) / pow(rv[i] - rv[i-1], 2)
                        + (FC(i+1) - FC(i)) / pow(rv[i+1] - rv[i], 2) );
        new_c[i] = ci/(bi - new_c[i-1]*ai);
        new_d[i] = (di - new_d[i-1]*ai)/(bi - new_c[i-1]*ai);
    }
    ai = 1.0/(rv[nr-1] - rv[nr-2]);
    bi = 2.0/(rv[nr-1] - rv[nr-2]);
    di = 3.0 * (FC(nr-1) - FC(nr-2)) / pow(rv[nr-1] - rv[nr-2], 2);
    new_d[nr-1] = (di - new_d[nr - 2]*ai)/(bi - new_c[nr - 2]*ai);
    out0[(size_t)(nr - 1) * stride] = new_d[nr - 1];
    for(int i = nr - 2; i >= 0; i--) out0[(size_t)i * stride] = new_d[i] - new_c[i] * out0[(size_t)(i+1) * stride];
    #undef FC
}

/* Set_Slopes_Multi: G2S_GlobalMultiDimSpline3D.cpp:313-431 */
static void set_slopes_multi_g(mdsg* S){
    int nr = S->nr, nt = S->nt, np = S->np;
    size_t ntot = (size_t)nr * nt * np, stride = (size_t)nt * np;
    double* new_c = (double*)malloc(sizeof(double) * (size_t)nr);
    double* new_d = (double*)malloc(sizeof(double) * (size_t)nr);
    double* dfdt = (double*)malloc(sizeof(double) * ntot);
    double* dfdp = (double*)malloc(sizeof(double) * ntot);
    for(int mt = 0; mt < nt; mt++) for(int mp = 0; mp < np; mp++)
        slopes_along_r(S->rv, nr, &GG(S, 0, mt, mp), stride, &GGS(S, fs, 0, mt, mp), new_c, new_d, 0);
    for(int mt = 0; mt < nt; mt++) for(int mp = 0; mp < np; mp++){
        int mt_up = IMIN(mt + 1, nt - 1), mt_dn = IMAX(mt - 1, 0);
        int mp_up = IMIN(mp + 1, np - 1), mp_dn = IMAX(mp - 1, 0);
        for(int mr = 0; mr < nr; mr++){
            dfdt[((size_t)mr * nt + mt) * np + mp] = (GG(S, mr, mt_up, mp) - GG(S, mr, mt_dn, mp))/(S->tv[mt_up] - S->tv[mt_dn]);
            dfdp[((size_t)mr * nt + mt) * np + mp] = (GG(S, mr, mt, mp_up) - GG(S, mr, mt, mp_dn))/(S->pv[mp_up] - S->pv[mp_dn]);
        }
    }
    for(int mt = 0; mt < nt; mt++) for(int mp = 0; mp < np; mp++)
        slopes_along_r(S->rv, nr, &dfdt[(size_t)mt * np + mp], stride, &GGS(S, fts, 0, mt, mp), new_c, new_d, 1);
    for(int mt = 0; mt < nt; mt++) for(int mp = 0; mp < np; mp++)
        slopes_along_r(S->rv, nr, &dfdp[(size_t)mt * np + mp], stride, &GGS(S, fps, 0, mt, mp), new_c, new_d, 1);
    free(new_c); free(new_d); free(dfdt); free(dfdp);
}

/* ---- vertical spline evaluations at node (kt, kp), segment kr: :481-565 ---- */
static double gv_f(double r, const mdsg* S, int kr, int kt, int kp){
    const double* rv = S->rv;
    double X = (r - rv[kr])/(rv[kr+1] - rv[kr]);
    double A = GGS(S, fs, kr, kt, kp) * (rv[kr+1] - rv[kr]) - (GG(S, kr+1, kt, kp) - GG(S, kr, kt, kp));
    double B = -GGS(S, fs, kr+1, kt, kp) * (rv[kr+1] - rv[kr]) + (GG(S, kr+1, kt, kp) - GG(S, kr, kt, kp));
    return (1.0 - X) * GG(S, kr, kt, kp) + X * GG(S, kr+1, kt, kp) + X * (1.0 - X) * (A * (1.0 - X ) + B * X);
}
static double gv_dfdr(double r, const mdsg* S, int kr, int kt, int kp){
    const double* rv = S->rv;
    double X = (r - rv[kr])/(rv[kr+1] - rv[kr]);
    double A = GGS(S, fs, kr, kt, kp) * (rv[kr+1] - rv[kr]) - (GG(S, kr+1, kt, kp) - GG(S, kr, kt, kp));
    double B = -GGS(S, fs, kr+1, kt, kp) * (rv[kr+1] - rv[kr]) + (GG(S, kr+1, kt, kp) - GG(S, kr, kt, kp));
    return (GG(S, kr+1, kt, kp) - GG(S, kr, kt, kp))/(rv[kr+1] - rv[kr])
         + (1.0 - 2.0 * X) * (A * (1.0 - X) + B * X)/(rv[kr+1] - rv[kr])
         + X * (1.0 - X) * (B - A)/(rv[kr+1] - rv[kr]);
}
static void gfd_t_nodes(const mdsg* S, int kr, int kt, int kp, double* a, double* b){
    int kt_up = IMIN(kt + 1, S->nt - 1), kt_dn = IMAX(kt - 1, 0);
    *a = (GG(S, kr, kt_up, kp) - GG(S, kr, kt_dn, kp))/(S->tv[kt_up] - S->tv[kt_dn]);
    *b = (GG(S, kr+1, kt_up, kp) - GG(S, kr+1, kt_dn, kp))/(S->tv[kt_up] - S->tv[kt_dn]);
}
static void gfd_p_nodes(const mdsg* S, int kr, int kt, int kp, double* a, double* b){
    int kp_up = IMIN(kp + 1, S->np - 1), kp_dn = IMAX(kp - 1, 0);
    *a = (GG(S, kr, kt, kp_up) - GG(S, kr, kt, kp_dn))/(S->pv[kp_up] - S->pv[kp_dn]);
    *b = (GG(S, kr+1, kt, kp_up) - GG(S, kr+1, kt, kp_dn))/(S->pv[kp_up] - S->pv[kp_dn]);
}
static double gv_dfdt(double r, const mdsg* S, int kr, int kt, int kp){
    const double* rv = S->rv;
    double dfdt_kr, dfdt_krp1; gfd_t_nodes(S, kr, kt, kp, &dfdt_kr, &dfdt_krp1);
    double X = (r - rv[kr])/(rv[kr+1] - rv[kr]);
    double A = GGS(S, fts, kr, kt, kp) * (rv[kr+1] - rv[kr]) - (dfdt_krp1 - dfdt_kr);
    double B = -GGS(S, fts, kr+1, kt, kp) * (rv[kr+1] - rv[kr]) + (dfdt_krp1 - dfdt_kr);
    return (1.0 - X) * dfdt_kr + X * dfdt_krp1 + X * (1.0 - X) * (A * (1.0 - X ) + B * X);
}
static double gv_dfdp(double r, const mdsg* S, int kr, int kt, int kp){
    const double* rv = S->rv;
    double dfdp_kr, dfdp_krp1; gfd_p_nodes(S, kr, kt, kp, &dfdp_kr, &dfdp_krp1);
    double X = (r - rv[kr])/(rv[kr+1] - rv[kr]);
    double A = GGS(S, fps, kr, kt, kp) * (rv[kr+1] - rv[kr]) - (dfdp_krp1 - dfdp_kr);
    double B = -GGS(S, fps, kr+1, kt, kp) * (rv[kr+1] - rv[kr]) + (dfdp_krp1 - dfdp_kr);
    return (1.0 - X) * dfdp_kr + X * dfdp_krp1 + X * (1.0 - X) * (A * (1.0 - X ) + B * X);
}
static double gv_ddfdrdr(double r, const mdsg* S, int kr, int kt, int kp){
    const double* rv = S->rv;
    double X = (r - rv[kr])/(rv[kr+1] - rv[kr]);
    double A = GGS(S, fs, kr, kt, kp) * (rv[kr+1] - rv[kr]) - (GG(S, kr+1, kt, kp) - GG(S, kr, kt, kp));
    double B = -GGS(S, fs, kr+1, kt, kp) * (rv[kr+1] - rv[kr]) + (GG(S, kr+1, kt, kp) - GG(S, kr, kt, kp));
    return 2.0 * (B - 2.0 * A + (A - B) * 3.0 * X)/pow(rv[kr+1] - rv[kr],2);
}
static double gv_ddfdrdt(double r, const mdsg* S, int kr, int kt, int kp){
    const double* rv = S->rv;
    double dfdt_kr, dfdt_krp1; gfd_t_nodes(S, kr, kt, kp, &dfdt_kr, &dfdt_krp1);
    double X = (r - rv[kr])/(rv[kr+1] - rv[kr]);
    double A = GGS(S, fts, kr, kt, kp) * (rv[kr+1] - rv[kr]) - (dfdt_krp1 - dfdt_kr);
    double B = -GGS(S, fts, kr+1, kt, kp) * (rv[kr+1] - rv[kr]) + (dfdt_krp1 - dfdt_kr);
    return (dfdt_krp1 - dfdt_krp1)/(rv[kr+1] - rv[kr])                                   /* Q12b: (x - x), :546 */
         + (1.0 - 2.0 * X) * (A * (1.0 - X) + B * X)/(rv[kr+1] - rv[kr])
         + X * (1.0 - X) * (B - A)/(rv[kr+1] - rv[kr]);
}
static double gv_ddfdrdp(double r, const mdsg* S, int kr, int kt, int kp){
    const double* rv = S->rv;
    double dfdp_kr, dfdp_krp1; gfd_p_nodes(S, kr, kt, kp, &dfdp_kr, &dfdp_krp1);
    double X = (r - rv[kr])/(rv[kr+1] - rv[kr]);
    double A = GGS(S, fps, kr, kt, kp) * (rv[kr+1] - rv[kr]) - (dfdp_krp1 - dfdp_kr);
    double B = -GGS(S, fps, kr+1, kt, kp) * (rv[kr+1] - rv[kr]) + (dfdp_krp1 - dfdp_kr);
    return (dfdp_krp1 - dfdp_krp1)/(rv[kr+1] - rv[kr])                                   /* Q12b, :562 */
         + (1.0 - 2.0 * X) * (A * (1.0 - X) + B * X)/(rv[kr+1] - rv[kr])
         + X * (1.0 - X) * (B - A)/(rv[kr+1] - rv[kr]);
}

/* ---- horizontal finite differences of vertical-spline values ("BiCubic_Deriv_*"): :571-749 ---- */
typedef double (*gevfn)(double, const mdsg*, int, int, int);
static double gfdt(gevfn g, double r, const mdsg* S, int kr, int kt, int kp){
    int up, dn; updn(kt, S->nt, &up, &dn);
    return (g(r, S, kr, up, kp) - g(r, S, kr, dn, kp))/(S->tv[up] - S->tv[dn]);
}
static double gfdp(gevfn g, double r, const mdsg* S, int kr, int kt, int kp){
    int up, dn; updn(kp, S->np, &up, &dn);
    return (g(r, S, kr, kt, up) - g(r, S, kr, kt, dn))/(S->pv[up] - S->pv[dn]);
}
static double gfdtp(gevfn g, double r, const mdsg* S, int kr, int kt, int kp){
    int tu, td, pu, pd; updn(kt, S->nt, &tu, &td); updn(kp, S->np, &pu, &pd);
    return (g(r, S, kr, tu, pu)
            - g(r, S, kr, tu, pd)
                - g(r, S, kr, td, pu)
                    + g(r, S, kr, td, pd))
                        /((S->tv[tu] - S->tv[td])*(S->pv[pu] - S->pv[pd]));
}
#define GBC_dfdt(r,S,k,i,j)          gfdt(gv_f, r, S, k, i, j)
#define GBC_dfdp(r,S,k,i,j)          gfdp(gv_f, r, S, k, i, j)
#define GBC_ddfdtdt(r,S,k,i,j)       gfdt(gv_dfdt, r, S, k, i, j)
#define GBC_ddfdpdp(r,S,k,i,j)       gfdp(gv_dfdp, r, S, k, i, j)
#define GBC_ddfdtdp(r,S,k,i,j)       gfdtp(gv_f, r, S, k, i, j)
#define GBC_dddfdrdrdt(r,S,k,i,j)    gfdt(gv_ddfdrdr, r, S, k, i, j)
#define GBC_dddfdrdrdp(r,S,k,i,j)    gfdp(gv_ddfdrdr, r, S, k, i, j)
#define GBC_dddfdrdtdp(r,S,k,i,j)    gfdtp(gv_dfdr, r, S, k, i, j)
#define GBC_dddfdtdtdp(r,S,k,i,j)    gfdtp(gv_dfdt, r, S, k, i, j)
#define GBC_dddfdtdpdp(r,S,k,i,j)    gfdtp(gv_dfdp, r, S, k, i, j)
#define GBC_ddddfdrdrdtdp(r,S,k,i,j) gfdtp(gv_ddfdrdr, r, S, k, i, j)

/* Eval_Spline_f: :755-807 (inputs already clamped by the callers) */
static double gg_eval_f(double r, double t, double p, mdsg* S){
    double A_vec[16], X_vec[16];
    int kr = find_segment(r, S->rv, S->nr, &S->accel[0]);
    int kt = find_segment(t, S->tv, S->nt, &S->accel[1]);
    int kp = find_segment(p, S->pv, S->np, &S->accel[2]);
    double dt_scalar = S->tv[kt+1] - S->tv[kt];
    double dp_scalar = S->pv[kp+1] - S->pv[kp];
    double t_scaled = (t - S->tv[kt])/dt_scalar;
    double p_scaled = (p - S->pv[kp])/dp_scalar;
    static const int ct[4] = {0, 1, 0, 1}, cp[4] = {0, 0, 1, 1};
    for(int q = 0; q < 4; q++){
        X_vec[q]      = gv_f(r, S, kr, kt + ct[q], kp + cp[q]);
        X_vec[4 + q]  = GBC_dfdt(r, S, kr, kt + ct[q], kp + cp[q])*dt_scalar;
        X_vec[8 + q]  = GBC_dfdp(r, S, kr, kt + ct[q], kp + cp[q])*dp_scalar;
        X_vec[12 + q] = GBC_ddfdtdp(r, S, kr, kt + ct[q], kp + cp[q])*dt_scalar*dp_scalar;
    }
    bic_matvec(X_vec, A_vec);
    return bic_poly(A_vec, t_scaled, p_scaled);
}

/* Eval_Spline_df: :809-897 (index 0: d/dr, 1: d/dt, 2: d/dp) */
static double gg_eval_df(double r, double t, double p, int index, mdsg* S){
    double A_vec[16], X_vec[16];
    int kr = find_segment(r, S->rv, S->nr, &S->accel[0]);
    int kt = find_segment(t, S->tv, S->nt, &S->accel[1]);
    int kp = find_segment(p, S->pv, S->np, &S->accel[2]);
    double dt_scalar = S->tv[kt+1] - S->tv[kt];
    double dp_scalar = S->pv[kp+1] - S->pv[kp];
    double t_scaled = (t - S->tv[kt])/dt_scalar;
    double p_scaled = (p - S->pv[kp])/dp_scalar;
    static const int ct[4] = {0, 1, 0, 1}, cp[4] = {0, 0, 1, 1};
    for(int q = 0; q < 4; q++){
        int i = kt + ct[q], j = kp + cp[q];
        if(index == 0){
            X_vec[q]      = gv_dfdr(r, S, kr, i, j);
            X_vec[4 + q]  = gv_ddfdrdt(r, S, kr, i, j)*dt_scalar;
            X_vec[8 + q]  = gv_ddfdrdp(r, S, kr, i, j)*dp_scalar;
            X_vec[12 + q] = GBC_dddfdrdtdp(r, S, kr, i, j)*dt_scalar*dp_scalar;
        } else if(index == 1){
            X_vec[q]      = GBC_dfdt(r, S, kr, i, j);
            X_vec[4 + q]  = GBC_ddfdtdt(r, S, kr, i, j)*dt_scalar;
            X_vec[8 + q]  = GBC_ddfdtdp(r, S, kr, i, j)*dp_scalar;
            X_vec[12 + q] = GBC_dddfdtdtdp(r, S, kr, i, j)*dt_scalar*dp_scalar;
        } else {
            X_vec[q]      = GBC_dfdp(r, S, kr, i, j);
            X_vec[4 + q]  = GBC_ddfdtdp(r, S, kr, i, j)*dt_scalar;
            X_vec[8 + q]  = GBC_ddfdpdp(r, S, kr, i, j)*dp_scalar;
            X_vec[12 + q] = GBC_dddfdtdpdp(r, S, kr, i, j)*dt_scalar*dp_scalar;
        }
    }
    bic_matvec(X_vec, A_vec);
    return bic_poly(A_vec, t_scaled, p_scaled);
}

/* Eval_Spline_AllOrder1 (order2 = 0, :1047-1222) and Eval_Spline_AllOrder2 (order2 = 1, :1224-1461).
 * out: f, dfdr, dfdt, dfdp [, ddfdrdr, ddfdtdt, ddfdpdp, ddfdrdt, ddfdrdp, ddfdtdp] */
static void gg_eval_all(const struct gridg* G, double r, double t, double p, mdsg* S, int order2, double* out){
    double A_vec[16], X_vec[16];
    double r_eval = DMIN(r, G->r_max);  r_eval = DMAX(r_eval, G->r_min);
    double t_eval = DMIN(t, G->t_max);  t_eval = DMAX(t_eval, G->t_min);
    double p_eval = DMIN(p, G->p_max);  p_eval = DMAX(p_eval, G->p_min);
    int kr = find_segment(r_eval, S->rv, S->nr, &S->accel[0]);
    int kt = find_segment(t_eval, S->tv, S->nt, &S->accel[1]);
    int kp = find_segment(p_eval, S->pv, S->np, &S->accel[2]);
    double dt_scalar = S->tv[kt+1] - S->tv[kt];
    double dp_scalar = S->pv[kp+1] - S->pv[kp];
    double t_scaled = (t_eval - S->tv[kt])/dt_scalar;
    double p_scaled = (p_eval - S->pv[kp])/dp_scalar;
    static const int ct[4] = {0, 1, 0, 1}, cp[4] = {0, 0, 1, 1};
    double Fdt[4], Fdp[4], Fdtp[4];
    for(int q = 0; q < 4; q++) Fdt[q]  = GBC_dfdt(r_eval, S, kr, kt + ct[q], kp + cp[q]);
    for(int q = 0; q < 4; q++) Fdp[q]  = GBC_dfdp(r_eval, S, kr, kt + ct[q], kp + cp[q]);
    for(int q = 0; q < 4; q++) Fdtp[q] = GBC_ddfdtdp(r_eval, S, kr, kt + ct[q], kp + cp[q]);

    /* f */
    for(int q = 0; q < 4; q++){
        X_vec[q] = gv_f(r_eval, S, kr, kt + ct[q], kp + cp[q]);
        X_vec[4 + q] = Fdt[q]*dt_scalar;
        X_vec[8 + q] = Fdp[q]*dp_scalar;
        X_vec[12 + q] = Fdtp[q]*dt_scalar*dp_scalar;
    }
    bic_matvec(X_vec, A_vec);
    out[0] = bic_poly(A_vec, t_scaled, p_scaled);

    /* df/dr (+ d2f/drdt, d2f/drdp, in SCALED coordinates: Q12c) */
    for(int q = 0; q < 4; q++){
        X_vec[q] = gv_dfdr(r_eval, S, kr, kt + ct[q], kp + cp[q]);
        X_vec[4 + q] = gv_ddfdrdt(r_eval, S, kr, kt + ct[q], kp + cp[q])*dt_scalar;
        X_vec[8 + q] = gv_ddfdrdp(r_eval, S, kr, kt + ct[q], kp + cp[q])*dp_scalar;
        X_vec[12 + q] = GBC_dddfdrdtdp(r_eval, S, kr, kt + ct[q], kp + cp[q])*dt_scalar*dp_scalar;
    }
    bic_matvec(X_vec, A_vec);
    out[1] = bic_poly(A_vec, t_scaled, p_scaled);
    if(order2){
        double ddfdrdt = 0;
        for(int k1 = 1; k1 < 4; k1++){
        for(int k2 = 0; k2 < 4; k2++){
            ddfdrdt+=1.0*k1*A_vec[k1 + 4*k2]*pow(t_scaled,k1-1)*pow(p_scaled,k2);
        }}
        double ddfdrdp = 0;
        for(int k1 = 0; k1 < 4; k1++){
        for(int k2 = 1; k2 < 4; k2++){
            ddfdrdp+=1.0*k2*A_vec[k1 + 4*k2]*pow(t_scaled,k1)*pow(p_scaled,k2-1);
        }}
        out[7] = ddfdrdt; out[8] = ddfdrdp;
    }

    /* df/dt (+ d2f/dt2, d2f/dtdp) */
    for(int q = 0; q < 4; q++){
        X_vec[q] = Fdt[q];
        X_vec[4 + q] = GBC_ddfdtdt(r_eval, S, kr, kt + ct[q], kp + cp[q])*dt_scalar;
        X_vec[8 + q] = Fdtp[q]*dp_scalar;
        X_vec[12 + q] = GBC_dddfdtdtdp(r_eval, S, kr, kt + ct[q], kp + cp[q])*dt_scalar*dp_scalar;
    }
    bic_matvec(X_vec, A_vec);
    out[2] = bic_poly(A_vec, t_scaled, p_scaled);
    if(order2){
        double ddfdtdt = 0;
        for(int k1 = 1; k1 < 4; k1++){
        for(int k2 = 0; k2 < 4; k2++){
            ddfdtdt+=1.0*k1*A_vec[k1 + 4*k2]*pow(t_scaled,k1-1)*pow(p_scaled,k2);
        }}
        double ddfdtdp = 0;
        for(int k1 = 0; k1 < 4; k1++){
        for(int k2 = 1; k2 < 4; k2++){
            ddfdtdp+=1.0*k2*A_vec[k1 + 4*k2]*pow(t_scaled,k1)*pow(p_scaled,k2-1);
        }}
        out[5] = ddfdtdt; out[9] = ddfdtdp;
    }

    /* df/dp (+ d2f/dp2) */
    for(int q = 0; q < 4; q++){
        X_vec[q] = Fdp[q];
        X_vec[4 + q] = Fdtp[q]*dt_scalar;
        X_vec[8 + q] = GBC_ddfdpdp(r_eval, S, kr, kt + ct[q], kp + cp[q])*dp_scalar;
        X_vec[12 + q] = GBC_dddfdtdpdp(r_eval, S, kr, kt + ct[q], kp + cp[q])*dt_scalar*dp_scalar;
    }
    bic_matvec(X_vec, A_vec);
    out[3] = bic_poly(A_vec, t_scaled, p_scaled);
    if(order2){
        double ddfdpdp = 0;
        for(int k1 = 0; k1 < 4; k1++){
        for(int k2 = 1; k2 < 4; k2++){
            ddfdpdp+=1.0*k2*A_vec[k1 + 4*k2]*pow(t_scaled,k1)*pow(p_scaled,k2-1);
        }}
        out[6] = ddfdpdp;

        /* d2f/dr2 */
        for(int q = 0; q < 4; q++){
            X_vec[q] = gv_ddfdrdr(r_eval, S, kr, kt + ct[q], kp + cp[q]);
            X_vec[4 + q] = GBC_dddfdrdrdt(r_eval, S, kr, kt + ct[q], kp + cp[q])*dt_scalar;
            X_vec[8 + q] = GBC_dddfdrdrdp(r_eval, S, kr, kt + ct[q], kp + cp[q])*dp_scalar;
            X_vec[12 + q] = GBC_ddddfdrdrdtdp(r_eval, S, kr, kt + ct[q], kp + cp[q])*dt_scalar*dp_scalar;
        }
        bic_matvec(X_vec, A_vec);
        out[4] = bic_poly(A_vec, t_scaled, p_scaled);
    }
}

/* ---- Atmo_State.h scalar API on the grid: :1502-1611 ---- */
static void gg_clamp(const struct gridg* G, double* r, double* t, double* p){
    double e;
    e = DMIN(*r, G->r_max); *r = DMAX(e, G->r_min);
    e = DMIN(*t, G->t_max); *t = DMAX(e, G->t_min);
    e = DMIN(*p, G->p_max); *p = DMAX(e, G->p_min);
}
static double gg_rho(struct gridg* G, double r, double t, double p){ gg_clamp(G, &r, &t, &p); return gg_eval_f(r, t, p, &G->Dens); }
static double gg_c(struct gridg* G, double r, double t, double p){ gg_clamp(G, &r, &t, &p); return sqrt(gamR * gg_eval_f(r, t, p, &G->Temp)); }
static double gg_u(struct gridg* G, double r, double t, double p){ gg_clamp(G, &r, &t, &p); return gg_eval_f(r, t, p, &G->Windu); }
static double gg_v(struct gridg* G, double r, double t, double p){ gg_clamp(G, &r, &t, &p); return gg_eval_f(r, t, p, &G->Windv); }
static double gg_c_diff(struct gridg* G, double r, double t, double p, int n){
    double re = r, te = t, pe = p; gg_clamp(G, &re, &te, &pe);
    return gamR / (2.0 * gg_c(G, r, t, p)) * gg_eval_df(re, te, pe, n, &G->Temp);
}
static double gg_u_diff(struct gridg* G, double r, double t, double p, int n){ gg_clamp(G, &r, &t, &p); return gg_eval_df(r, t, p, n, &G->Windu); }
static double gg_v_diff(struct gridg* G, double r, double t, double p, int n){ gg_clamp(G, &r, &t, &p); return gg_eval_df(r, t, p, n, &G->Windv); }

/* ---- loading: SetUp_G2S_Arrays + Load_G2S_Multi + Spline_Multi_G2S (:110-195, :1473-1492) ---- */
static void gg_free(struct gridg* G){
    if(!G) return;
    mdsg* S[4] = { &G->Temp, &G->Windu, &G->Windv, &G->Dens };
    for(int i = 0; i < 4; i++){ free(S[i]->f); free(S[i]->fs); free(S[i]->fts); free(S[i]->fps); }
    free(G->rv); free(G->tv); free(G->pv); free(G);
}
/* z_grnd_at_load: value of the z_grnd global when the profiles are read (taper width 0.2 as in the stratified Global loader) */
static struct gridg* gg_load(const char* prefix, const char* loclat, const char* loclon, const char* format, double z_grnd_at_load, double r_earth){
    int nt = count_newlines(loclat), np = count_newlines(loclon);
    char buf[512];
    snprintf(buf, sizeof buf, "%s%i.met", prefix, 0);
    int nr = count_newlines(buf);
    if(nt < 2 || np < 2 || nr < 3) return NULL;
    int fmt;
    if(strncmp(format, "zTuvdp", 6) == 0) fmt = 0; else if(strncmp(format, "zuvwTdp", 7) == 0) fmt = 1; else return NULL;
    struct gridg* G = (struct gridg*)calloc(1, sizeof(struct gridg));
    G->nr = nr; G->nt = nt; G->np = np;
    G->rv = malloc(sizeof(double) * (size_t)nr); G->tv = malloc(sizeof(double) * (size_t)nt); G->pv = malloc(sizeof(double) * (size_t)np);
    size_t ntot = (size_t)nr * nt * np;
    mdsg* S[4] = { &G->Temp, &G->Windu, &G->Windv, &G->Dens };
    for(int i = 0; i < 4; i++){
        S[i]->nr = nr; S[i]->nt = nt; S[i]->np = np; S[i]->accel[0] = S[i]->accel[1] = S[i]->accel[2] = 0;
        S[i]->rv = G->rv; S[i]->tv = G->tv; S[i]->pv = G->pv;
        S[i]->f = malloc(sizeof(double) * ntot); S[i]->fs = malloc(sizeof(double) * ntot);
        S[i]->fts = malloc(sizeof(double) * ntot); S[i]->fps = malloc(sizeof(double) * ntot);
    }
    FILE* fp = fopen(loclat, "r");
    for(int i = 0; i < nt; i++){ if(fscanf(fp, "%lf", &G->tv[i]) != 1) G->tv[i] = 0; G->tv[i] *= Pi/180.0; }
    fclose(fp);
    fp = fopen(loclon, "r");
    for(int i = 0; i < np; i++){ if(fscanf(fp, "%lf", &G->pv[i]) != 1) G->pv[i] = 0; G->pv[i] *= Pi/180.0; }
    fclose(fp);
    for(int ip = 0; ip < np; ip++) for(int it = 0; it < nt; it++){
        snprintf(buf, sizeof buf, "%s%i.met", prefix, it * np + ip);
        fp = fopen(buf, "r");
        if(!fp){ gg_free(G); return NULL; }
        for(int ir = 0; ir < nr; ir++){
            double tk[7] = {0,0,0,0,0,0,0}; int ntok = fmt ? 7 : 6;
            for(int j = 0; j < ntok; j++) if(fscanf(fp, "%lf", &tk[j]) != 1) tk[j] = 0.0;
            double zz, TT, uu, vv, rr;
            if(fmt == 0){ zz = tk[0]; TT = tk[1]; uu = tk[2]; vv = tk[3]; rr = tk[4]; }
            else { zz = tk[0]; uu = tk[1]; vv = tk[2]; TT = tk[4]; rr = tk[5]; }
            G->rv[ir] = zz;
            G->rv[ir] += r_earth;
            uu *= (2.0 / (1.0 + exp(-(G->rv[ir] - r_earth - z_grnd_at_load)/0.2)) - 1.0) / 1000.0;
            vv *= (2.0 / (1.0 + exp(-(G->rv[ir] - r_earth - z_grnd_at_load)/0.2)) - 1.0) / 1000.0;
            GG(&G->Temp, ir, it, ip) = TT; GG(&G->Windu, ir, it, ip) = uu; GG(&G->Windv, ir, it, ip) = vv; GG(&G->Dens, ir, it, ip) = rr;
        }
        fclose(fp);
    }
    G->r_min = G->rv[0]; G->r_max = G->rv[nr - 1];
    G->t_min = G->tv[0]; G->t_max = G->tv[nt - 1];
    G->p_min = G->pv[0]; G->p_max = G->pv[np - 1];
    set_slopes_multi_g(&G->Temp); set_slopes_multi_g(&G->Windu);
    set_slopes_multi_g(&G->Dens); set_slopes_multi_g(&G->Windv);
    return G;
}

/* ------------------------------------------------------------------------------------------ */
/* range-dependent spherical set: GeoAc.EquationSets.GlobalRngDep.cpp (sources struct = the Global one, c->G) */
/* ------------------------------------------------------------------------------------------ */
/* GeoAc_SetInitialConditions: GlobalRngDep.cpp:77-137.  r0 = altitude, theta0/phi0 = lat/lon [rad] */
static void grd_set_ic(orc_ctx* c, double r0, double theta0, double phi0){
    src_global* S = &c->G; double* y = ROW(c, 0); struct gridg* G = c->GG;
    double re = c->r_earth;
    S->src_loc[0] = r0 + re; S->src_loc[1] = theta0; S->src_loc[2] = phi0;
    S->c0 = gg_c(G, r0 + re, theta0, phi0);
    double MachComps[3] = { 0.0/S->c0, gg_v(G, r0 + re, theta0, phi0)/S->c0, gg_u(G, r0 + re, theta0, phi0)/S->c0 };
    double th = c->theta, ph = c->phi;
    double nu0[3]    = { sin(th),  cos(th)*sin(ph),  cos(th)*cos(ph) };
    double mu0_lt[3] = { cos(th), -sin(th)*sin(ph), -sin(th)*cos(ph) };
    double mu0_lp[3] = { 0.0,      cos(th)*cos(ph), -cos(th)*sin(ph) };
    double MachScalar = 1.0 + (nu0[0]*MachComps[0] + nu0[1]*MachComps[1] + nu0[2]*MachComps[2]);
    S->nu0 = 1.0/MachScalar;
    for(int i = 0; i < c->EqCnt; i++){
        if(i == 0) y[i] = r0 + re;
        else if(i == 1) y[i] = theta0;
        else if(i == 2) y[i] = phi0;
        else if(i < 6) y[i] = nu0[i-3]/MachScalar;
        else if(i < 9 || (i >= 12 && i < 15)) y[i] = 0.0;
        else if(i < 12) y[i] = mu0_lt[i-9]/MachScalar - nu0[i-9]/pow(MachScalar,2.0) * (mu0_lt[0]*MachComps[0] + mu0_lt[1]*MachComps[1] + mu0_lt[2]*MachComps[2]);
        else y[i] = mu0_lp[i-15]/MachScalar - nu0[i-15]/pow(MachScalar,2.0) * (mu0_lp[0]*MachComps[0] + mu0_lp[1]*MachComps[1] + mu0_lp[2]*MachComps[2]);
    }
}

/* geometric coefficients / terms shared by both branches of GeoAc_UpdateSources (:258-269, :346-364) */
static void grd_geo(src_global* S, double r, double theta, const double* nu){
    S->GeoCoeff[0] = 1.0;
    S->GeoCoeff[1] = 1.0/r;
    S->GeoCoeff[2] = 1.0/(r*cos(theta));
    S->GeoTerms[0] = 0.0;
    S->GeoTerms[1] = (nu[0]*S->v - nu[1]*S->w);
    S->GeoTerms[2] = (nu[0]*S->u - nu[2]*S->w)*cos(theta) + (nu[1]*S->u - nu[2]*S->v)*sin(theta);
    S->GeoTerms[0] += 1.0/r * (nu[1]*S->c_gr[1] + nu[2]*S->c_gr[2]);
    S->GeoTerms[1] += -nu[0]*S->c_gr[1] + nu[2]*S->c_gr[2]*tan(theta);
    S->GeoTerms[2] += -S->c_gr[2]*(nu[0]*cos(theta) + nu[1]*sin(theta));
}

/* GeoAc_UpdateSources: GlobalRngDep.cpp:226-386 */
static void grd_update_sources(orc_ctx* c, const double* cur){
    src_global* S = &c->G; struct gridg* G = c->GG;
    double r = cur[0], theta = cur[1], phi = cur[2];
    double nu[3] = { cur[3], cur[4], cur[5] };
    double oT[10], oU[10], oV[10];
    if(!c->CalcAmp){
        gg_eval_all(G, r, theta, phi, &G->Temp, 0, oT);
        for(int n = 0; n < 3; n++){ G->Windu.accel[n] = G->Temp.accel[n]; G->Windv.accel[n] = G->Temp.accel[n]; }
        gg_eval_all(G, r, theta, phi, &G->Windu, 0, oU);
        gg_eval_all(G, r, theta, phi, &G->Windv, 0, oV);
        S->u = oU[0]; S->v = oV[0]; S->w = 0.0;
        for(int n = 0; n < 3; n++){ S->du[n] = oU[1+n]; S->dv[n] = oV[1+n]; }
        S->c = sqrt(gamR * oT[0]);
        for(int n = 0; n < 3; n++){ S->dc[n] = gamR / (2.0 * S->c) * oT[1+n]; S->dw[n] = 0.0; }
        S->nu_mag = sqrt(nu[0]*nu[0] + nu[1]*nu[1] + nu[2]*nu[2]);
        S->c_gr[0] = S->c*nu[0]/S->nu_mag + S->w;
        S->c_gr[1] = S->c*nu[1]/S->nu_mag + S->v;
        S->c_gr[2] = S->c*nu[2]/S->nu_mag + S->u;
        S->c_gr_mag = sqrt(pow(S->c_gr[0],2) + pow(S->c_gr[1],2) + pow(S->c_gr[2],2));
        grd_geo(S, r, theta, nu);
        return;
    }
    double Rl[2][3] = { { cur[6],  cur[7],  cur[8]  }, { cur[12], cur[13], cur[14] } };
    double ml[2][3] = { { cur[9],  cur[10], cur[11] }, { cur[15], cur[16], cur[17] } };
    double dtemp[3], ddtemp[3][3], ddWindu[3][3], ddWindv[3][3];
    gg_eval_all(G, r, theta, phi, &G->Temp, 1, oT);
    for(int n = 0; n < 3; n++){ G->Windu.accel[n] = G->Temp.accel[n]; G->Windv.accel[n] = G->Temp.accel[n]; }
    gg_eval_all(G, r, theta, phi, &G->Windu, 1, oU);
    gg_eval_all(G, r, theta, phi, &G->Windv, 1, oV);
    S->u = oU[0]; S->v = oV[0]; S->w = 0.0;
    for(int n = 0; n < 3; n++){ dtemp[n] = oT[1+n]; S->du[n] = oU[1+n]; S->dv[n] = oV[1+n]; }
    /* out: [4] rr, [5] tt, [6] pp, [7] rt, [8] rp, [9] tp */
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
            for(int a = 0; a < 2; a++)
                S->ddc[n][a] += Rl[a][m]*(gamR/(2.0*S->c) * ddtemp[n][m] - pow(gamR,2)/(4.0 * pow(S->c,3)) * dtemp[n]*dtemp[m]);
            for(int a = 0; a < 2; a++) S->ddu[n][a] += Rl[a][m]*ddWindu[n][m];
            for(int a = 0; a < 2; a++) S->ddv[n][a] += Rl[a][m]*ddWindv[n][m];
            for(int a = 0; a < 2; a++) S->ddw[n][a] += Rl[a][m]*0.0;
        }
    }
    for(int a = 0; a < 2; a++){ S->dc[3+a] = 0.0; S->du[3+a] = 0.0; S->dv[3+a] = 0.0; S->dw[3+a] = 0.0; }
    for(int n = 0; n < 3; n++){
        for(int a = 0; a < 2; a++) S->dc[3+a] += Rl[a][n]*S->dc[n];
        for(int a = 0; a < 2; a++) S->du[3+a] += Rl[a][n]*S->du[n];
        for(int a = 0; a < 2; a++) S->dv[3+a] += Rl[a][n]*S->dv[n];
        for(int a = 0; a < 2; a++) S->dw[3+a] += Rl[a][n]*S->dw[n];
    }
    S->nu_mag = sqrt(nu[0]*nu[0] + nu[1]*nu[1] + nu[2]*nu[2]);
    for(int a = 0; a < 2; a++) S->dnu_mag[a] = (nu[0]*ml[a][0] + nu[1]*ml[a][1] + nu[2]*ml[a][2])/S->nu_mag;
    S->c_gr[0] = S->c*nu[0]/S->nu_mag + S->w;
    S->c_gr[1] = S->c*nu[1]/S->nu_mag + S->v;
    S->c_gr[2] = S->c*nu[2]/S->nu_mag + S->u;
    S->c_gr_mag = sqrt(pow(S->c_gr[0],2) + pow(S->c_gr[1],2) + pow(S->c_gr[2],2));
    for(int a = 0; a < 2; a++){
        double wind_d[3] = { S->dw[3+a], S->dv[3+a], S->du[3+a] };
        for(int i = 0; i < 3; i++)
            S->dc_gr[i][a] = nu[i]/S->nu_mag*S->dc[3+a] + S->c*ml[a][i]/S->nu_mag - S->c*nu[i]/pow(S->nu_mag,2) * S->dnu_mag[a] + wind_d[i];
        S->dc_gr_mag[a] = (S->c_gr[0]*S->dc_gr[0][a] + S->c_gr[1]*S->dc_gr[1][a] + S->c_gr[2]*S->dc_gr[2][a])/S->c_gr_mag;
    }
    grd_geo(S, r, theta, nu);
    for(int a = 0; a < 2; a++){
        const double* R_l = Rl[a]; const double* mu_l = ml[a];
        S->d_GeoCoeff[0][a] = 0.0;
        S->d_GeoCoeff[1][a] = -R_l[0]/(pow(r,2));
        S->d_GeoCoeff[2][a] = -R_l[0]/(pow(r,2)*cos(theta)) + sin(theta)*R_l[1]/(r*pow(cos(theta),2));
        double dwa = S->dw[3+a], dva = S->dv[3+a], dua = S->du[3+a];
        S->d_GeoTerms[0][a] = 0.0;
        S->d_GeoTerms[1][a] = (mu_l[0]*S->v + nu[0]*dva - mu_l[1]*S->w - nu[1] * dwa);
        S->d_GeoTerms[2][a] = (mu_l[0]*S->u + nu[0]*dua - mu_l[2]*S->w - nu[2] * dwa)*cos(theta) - (nu[0]*S->u - nu[2]*S->w)*R_l[1]*sin(theta)
                            + (mu_l[1]*S->u + nu[1]*dua - mu_l[2]*S->v - nu[2] * dva)*sin(theta) + (nu[1]*S->u - nu[2]*S->v)*R_l[1]*cos(theta);
        S->d_GeoTerms[0][a] += -R_l[0]/pow(r,2)*(nu[1]*S->c_gr[1] + nu[2]*S->c_gr[2])
                             + 1.0/r*(mu_l[1]*S->c_gr[1] + nu[1]*S->dc_gr[1][a] + mu_l[2]*S->c_gr[2] + nu[2]*S->dc_gr[2][a]);
        S->d_GeoTerms[1][a] += -mu_l[0]*S->c_gr[1] - nu[0]*S->dc_gr[1][a] + mu_l[2]*S->c_gr[2]*tan(theta) + nu[2]*S->dc_gr[2][a]*tan(theta) + nu[2]*S->c_gr[2]*R_l[1]/pow(cos(theta),2);
        S->d_GeoTerms[2][a] += -S->dc_gr[2][a]*(nu[0]*cos(theta) + nu[1]*sin(theta)) - S->c_gr[2]*(mu_l[0]*cos(theta) - nu[0]*R_l[1]*sin(theta) + mu_l[1]*sin(theta) + nu[1]*R_l[1]*cos(theta));
    }
}

/* GeoAc_BreakCheck: GlobalRngDep.cpp:523-535 (lat/lon box; limits in whatever unit they were assigned) */
static int grd_break_check(const orc_ctx* c, int k){
    const double* y = ROW(c, k);
    int check = 0;
    double alt = y[0], lat = y[1], lon = y[2];
    if(alt > c->vert_limit) check = 1;
    if(lat < c->x_min_limit || lat > c->x_max_limit) check = 1;
    if(lon < c->y_min_limit || lon > c->y_max_limit) check = 1;
    return check;
}

/* travel-time / attenuation segment: GlobalRngDep.cpp:549-612, 657-693 */
static double grd_tt_seg(orc_ctx* c, int n){
    struct gridg* G = c->GG;
    const double* a = ROW(c, n); const double* b = ROW(c, n+1);
    double dr = b[0] - a[0], dt = b[1] - a[1], dp = b[2] - a[2];
    double r = a[0] + dr/2.0, t = a[1] + dt/2.0, p = a[2] + dp/2.0;
    double ds = sqrt(pow(dr,2) + pow(r*dt,2) + pow(r*cos(t)*dp,2));
    double nu[3];
    nu[0] = a[3] + (b[3] - a[3])/2.0;
    nu[1] = a[4] + (b[4] - a[4])/2.0;
    nu[2] = a[5] + (b[5] - a[5])/2.0;
    double nu_mag = sqrt(nu[0]*nu[0] + nu[1]*nu[1] + nu[2]*nu[2]);
    double cc = gg_c(G, r, t, p), vv = gg_v(G, r, t, p), uu = gg_u(G, r, t, p);
    double c_prop[3] = { cc*nu[0]/nu_mag + 0.0, cc*nu[1]/nu_mag + vv, cc*nu[2]/nu_mag + uu };
    double c_prop_mag = sqrt(pow(c_prop[0],2) + pow(c_prop[1],2) + pow(c_prop[2],2));
    return ds/c_prop_mag;
}
static double grd_att_seg(orc_ctx* c, int n, double freq){
    struct gridg* G = c->GG;
    const double* a = ROW(c, n); const double* b = ROW(c, n+1);
    double dr = b[0] - a[0], dt = b[1] - a[1], dp = b[2] - a[2];
    double r = a[0] + dr/2.0, t = a[1] + dt/2.0, p = a[2] + dp/2.0;
    double ds = sqrt(pow(dr,2) + pow(r*dt,2) + pow(r*sin(t)*dp,2));
    /* SuthBass_Alpha(r, theta, phi, f) (Atmo_State.Absorption.Global.cpp:12-141): reference state at radius z_grnd (clamped to
     * r_min) and the LOCAL latitude / longitude */
    double c_g = gg_c(G, c->z_grnd, t, p), rho_g = gg_rho(G, c->z_grnd, t, p);
    double c_z = gg_c(G, r, t, p), rho_z = gg_rho(G, r, t, p);
    return suthbass_core(c, r - c->r_earth, c_g, rho_g, c_z, rho_z, freq)*ds;
}

/* GeoAc_Jacobian / GeoAc_Amplitude: GlobalRngDep.cpp:617-652 (Q3, Q4 as in the stratified set) */
static double grd_jacobian(orc_ctx* c, int k){
    struct gridg* G = c->GG; const double* y = ROW(c, k);
    double r = y[0], theta = y[1], phi = y[2];
    double nu[3] = { y[3], y[4], y[5] };
    double nu_mag = sqrt(nu[0]*nu[0] + nu[1]*nu[1] + nu[2]*nu[2]);
    double cc = gg_c(G, r, theta, phi), vv = gg_v(G, r, theta, phi), uu = gg_u(G, r, theta, phi);
    double c_prop[3] = { cc*nu[0]/nu_mag + 0.0, cc*nu[1]/nu_mag + vv, cc*nu[2]/nu_mag + uu };
    double c_prop_mag = sqrt(pow(c_prop[0],2) + pow(c_prop[1],2) + pow(c_prop[2],2));
    double dr_ds = c_prop[0]/c_prop_mag, dt_ds = 1.0/r*c_prop[1]/c_prop_mag, dp_ds = 1.0/(r*sin(theta))*c_prop[2]/c_prop_mag;
    double dr_dlt = y[6],  dt_dlt = y[7],  dp_dlt = y[8];
    double dr_dlp = y[12], dt_dlp = y[13], dp_dlp = y[14];
    return pow(r,2)*cos(theta)*(dr_ds*(dt_dlt*dp_dlp - dt_dlp*dp_dlt) - dr_dlt*(dt_ds*dp_dlp - dp_ds*dt_dlp) + dr_dlp*(dt_ds*dp_dlt - dp_ds*dt_dlt));
}
static double grd_amplitude(orc_ctx* c, int k){
    const src_global* S = &c->G; struct gridg* G = c->GG; const double* y = ROW(c, k);
    double r0 = S->src_loc[0], theta0 = S->src_loc[1], phi0 = S->src_loc[2];
    double r = y[0], theta = y[1], phi = y[2];
    double nu[3] = { y[3], y[4], y[5] };
    double th = c->theta, ph = c->phi;
    double nu0[3] = { sin(th), cos(th)*sin(ph), cos(th)*cos(ph) };
    double cc = gg_c(G, r, theta, phi), vv = gg_v(G, r, theta, phi), uu = gg_u(G, r, theta, phi);
    double nu_mag = (S->c0 - nu[0]*0.0 - nu[1]*vv - nu[2]*uu)/cc;
    double nu_mag0 = S->nu0;
    double c_prop[3]  = { cc*nu[0]/nu_mag + 0.0, cc*nu[1]/nu_mag + vv, cc*nu[2]/nu_mag + uu };
    double c_prop0[3] = { S->c0*nu0[0]/nu_mag0 + 0.0, S->c0*nu0[1]/nu_mag + gg_v(G, r0, theta0, phi0), S->c0*nu0[2]/nu_mag + gg_u(G, r0, theta0, phi0) };
    double c_prop_mag  = sqrt(pow(c_prop[0],2) +  pow(c_prop[1],2) +  pow(c_prop[2],2));
    double c_prop_mag0 = sqrt(pow(c_prop0[0],2) + pow(c_prop0[1],2) + pow(c_prop0[2],2));
    double D = grd_jacobian(c, k);
    double Amp_Num = gg_rho(G, r, theta, phi) * nu_mag * pow(gg_c(G, r, theta, phi),3) * c_prop_mag0 * cos(th);
    double Amp_Den = gg_rho(G, r0, theta0, phi0)* nu_mag0* pow(gg_c(G, r0, theta0, phi0),3)* c_prop_mag  * D;
    return 1.0/(4.0*Pi)*sqrt(fabs(Amp_Num/Amp_Den));
}

/* ApproximateIntercept + SetReflectionConditions: GlobalRngDep.cpp:141-210 (linear intercept only: Q1, :147-148) */
static void grd_reflect(orc_ctx* c, int k){
    const src_global* S = &c->G; struct gridg* G = c->GG;
    double prev[18];
    const double* yk = ROW(c, k); const double* ykm = ROW(c, k-1);
    double rg = c->r_earth + c->z_grnd;
    double dr_k = yk[0] - ykm[0];
    double dr_grnd = ykm[0] - rg;
    for(int i = 0; i < c->EqCnt; i++) prev[i] = ykm[i] + (ykm[i] - yk[i])/dr_k*dr_grnd;

    double c_ref = gg_c(G, prev[0], prev[1], prev[2]);
    double dnu_r_ds = - 1.0/c_ref * (S->c0/c_ref * gg_c_diff(G, prev[0], prev[1], prev[2], 0)
                                     + prev[3] * 0.0
                                     + prev[4] * gg_v_diff(G, prev[0], prev[1], prev[2], 0)
                                     + prev[5] * gg_u_diff(G, prev[0], prev[1], prev[2], 0)
                                     + c_ref/prev[0] * (pow(prev[4],2) + pow(prev[5],2)));
    double* y0 = ROW(c, 0);
    for(int i = 0; i < c->EqCnt; i++){
        if(i == 0) y0[i] = rg;
        else if(i == 3 || i == 6 || i == 12) y0[i] = -prev[i];
        else if(i == 9 || i == 15) y0[i] = -prev[i] + 2.0*dnu_r_ds * prev[i-3]/( c_ref / S->c0 * prev[3]);
        else y0[i] = prev[i];
    }
}
