/* TEST INFRASTRUCTURE ONLY (see geoac_oracle.h).  Plain-C CPU restatement of GeoAc's ray-fan hot
 * path.  It keeps the reference's arithmetic (operand order, pow() calls, clamps, the hinted
 * segment search with its per-spline cursor, the stray-semicolon intercept of the Global set, ...)
 * so that, built with the same compiler and libm, it reproduces the compiled reference bit for bit;
 * it drops only the redundancy (one segment search per distinct abscissa instead of ~40).
 *
 * Reference files restated here (paths under /root/reference/Code):
 *   GeoAc/GeoAc.Solver.cpp, GeoAc/GeoAc.EquationSets.{2DStratified,3DStratified,Global}.cpp,
 *   Atmo/G2S_Spline1D.cpp, Atmo/G2S_GlobalSpline1D.cpp, Atmo/Atmo_State.Absorption{,.Global}.cpp,
 *   GeoAc/GeoAc.Parameters{,.Global}.cpp, GeoAc/GeoAc.Interface{,.Global}.cpp (EqCnt table),
 *   the fan / bounce / post-pass loops of GeoAc2D_main.cpp, GeoAc3D_main.cpp, GeoAcGlobal_main.cpp.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "geoac_oracle.h"

/* GeoAc.Parameters.cpp:27-31 */
static const double Pi   = 3.141592653589793238462643;
static const double gam  = 1.4;
static const double Rgas = 287.05;
/* G2S_Spline1D.cpp:332 / G2S_GlobalSpline1D.cpp:343 */
static const double gamR = 0.00040187;

#define DMIN(a,b) (((b) < (a)) ? (b) : (a))      /* std::min */
#define DMAX(a,b) (((a) < (b)) ? (b) : (a))      /* std::max */

/* ------------------------------------------------------------------------------------------ */
/* 1-D natural cubic spline: G2S_Spline1D.h:43-49                                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int     length;
    int     accel;      /* cursor of the hinted search; history matters only at exact node ties (Q13) */
    double* x_vals;
    double* f_vals;
    double* slopes;
} spline1d;

/* Set_Slopes: G2S_Spline1D.cpp:161-196 / G2S_GlobalSpline1D.cpp:175-210 (Thomas algorithm, natural BCs) */
static void set_slopes(spline1d* S){
    int n = S->length;
    double* new_c = (double*)malloc(sizeof(double) * (size_t)n);
    double* new_d = (double*)malloc(sizeof(double) * (size_t)n);
    const double* x = S->x_vals; const double* f = S->f_vals;
    double ai, bi, ci, di;

    bi = 2.0 / (x[1] - x[0]);
    ci = 1.0 / (x[1] - x[0]);
    di = 3.0 * (f[1] - f[0]) / pow(x[1] - x[0], 2);
    new_c[0] = ci/bi;
    new_d[0] = di/bi;

    for(int i = 1; i < n - 1; i++){
        ai = 1.0/(x[i] - x[i-1]);
        bi = 2.0 * (1.0/(x[i] - x[i-1]) + 1.0/(x[i+1] - x[i]));
        ci = 1.0/(x[i+1] - x[i]);
        di = 3.0 * ((f[i] - f[i-1]) / pow(x[i] - x[i-1], 2)
                    + (f[i+1] - f[i]) / pow(x[i+1] - x[i], 2) );
        new_c[i] = ci/(bi - new_c[i-1]*ai);
        new_d[i] = (di - new_d[i-1]*ai)/(bi - new_c[i-1]*ai);
    }
    ai = 1.0/(x[n-1] - x[n-2]);
    bi = 2.0/(x[n-1] - x[n-2]);
    di = 3.0 * (f[n-1] - f[n-2]) / pow(x[n-1] - x[n-2], 2);
    new_d[n-1] = (di - new_d[n-2]*ai)/(bi - new_c[n-2]*ai);

    S->slopes[n-1] = new_d[n-1];
    for(int i = n - 2; i > -1; i--) S->slopes[i] = new_d[i] - new_c[i] * S->slopes[i+1];
    free(new_c); free(new_d);
}

/* Find_Segment: G2S_Spline1D.cpp:202-243 / G2S_GlobalSpline1D.cpp:216-257 */
static int find_segment(double x, const double* xv, int length, int* prev){
    int index = length + 1;
    int done = 0;
    if(x > xv[length-1] || x < xv[0]) return length + 1;   /* reference prints and falls off the end (UB); callers clamp, never reached */
    if(x >= xv[*prev] && x <= xv[*prev+1]) done = 1;
    if(!done && *prev+2 <= length-1){
        if(x >= xv[*prev+1] && x <= xv[*prev+2]){ done = 1; *prev = *prev + 1; }
    }
    if(!done && *prev-1 >= 0){
        if(x >= xv[*prev-1] && x <= xv[*prev]){ done = 1; *prev = *prev - 1; }
    }
    if(!done){
        for(int i = 0; i < length; i++){
            if(x >= xv[i] && x <= xv[i+1]){ index = i; break; }
            if(x >= xv[length-2-i] && x < xv[length-1-i]){ index = (length - 2) - i; break; }
        }
        *prev = index;
    }
    return *prev;
}

/* Eval_Spline_f / df / ddf: G2S_Spline1D.cpp:245-281 / G2S_GlobalSpline1D.cpp:259-295 */
static double eval_f(double x, spline1d* S){
    int k = find_segment(x, S->x_vals, S->length, &S->accel);
    if(k < S->length){
        const double* xv = S->x_vals; const double* fv = S->f_vals; const double* sl = S->slopes;
        double X = (x - xv[k])/(xv[k+1] - xv[k]);
        double A = sl[k] * (xv[k+1] - xv[k]) - (fv[k+1] - fv[k]);
        double B = -sl[k+1] * (xv[k+1] - xv[k]) + (fv[k+1] - fv[k]);
        return (1.0 - X) * fv[k] + X * fv[k+1] + X * (1.0 - X) * (A * (1.0 - X ) + B * X);
    }
    return 0.0;
}
static double eval_df(double x, spline1d* S){
    int k = find_segment(x, S->x_vals, S->length, &S->accel);
    if(k < S->length){
        const double* xv = S->x_vals; const double* fv = S->f_vals; const double* sl = S->slopes;
        double X = (x - xv[k])/(xv[k+1] - xv[k]);
        double A = sl[k] * (xv[k+1] - xv[k]) - (fv[k+1] - fv[k]);
        double B = -sl[k+1] * (xv[k+1] - xv[k]) + (fv[k+1] - fv[k]);
        return (fv[k+1] - fv[k])/(xv[k+1] - xv[k])
             + (1.0 - 2.0 * X) * (A * (1.0 - X) + B * X)/(xv[k+1] - xv[k])
             + X * (1.0 - X) * (B - A)/(xv[k+1] - xv[k]);
    }
    return 0.0;
}
static double eval_ddf(double x, spline1d* S){
    int k = find_segment(x, S->x_vals, S->length, &S->accel);
    if(k < S->length){
        const double* xv = S->x_vals; const double* fv = S->f_vals; const double* sl = S->slopes;
        double X = (x - xv[k])/(xv[k+1] - xv[k]);
        double A = sl[k] * (xv[k+1] - xv[k]) - (fv[k+1] - fv[k]);
        double B = -sl[k+1] * (xv[k+1] - xv[k]) + (fv[k+1] - fv[k]);
        return 2.0 * (B - 2.0 * A + (A - B) * 3.0 * X)/pow(xv[k+1] - xv[k],2);
    }
    return 0.0;
}

/* ------------------------------------------------------------------------------------------ */
/* context = the reference's process-wide globals, made re-entrant                              */
/* ------------------------------------------------------------------------------------------ */
typedef struct {          /* GeoAc_Sources_Struct of the Global set: EquationSets.Global.cpp:24-59 */
    double src_loc[3], c0;
    double c, dc[5], ddc[3][2];
    double w, dw[5], ddw[3][2];
    double v, dv[5], ddv[3][2];
    double u, du[5], ddu[3][2];
    double nu0, nu_mag, dnu_mag[2];
    double c_gr[3], c_gr_mag, dc_gr[3][2], dc_gr_mag[2];
    double GeoCoeff[3], d_GeoCoeff[3][2];
    double GeoTerms[3], d_GeoTerms[3][2];
} src_global;

typedef struct {          /* EquationSets.3DStratified.cpp:23-54 */
    double src_loc[3], c0;
    double nu0_xy[2], mu0_xy[2][2];
    double c, dc, ddc, u, du, ddu, v, dv, ddv, w, dw, ddw;
    double nu_mag, dnu_mag[2];
    double c_prop[3], c_prop_mag, dc_prop[3][2], dc_prop_mag[2];
} src_3d;

typedef struct {          /* EquationSets.3DRngDep.cpp:25-55 */
    double src_loc[3], c0;
    double c, dc[5], ddc[3][2];
    double u, du[5], ddu[3][2];
    double v, dv[5], ddv[3][2];
    double w, dw[5], ddw[3][2];
    double nu0, nu_mag, dnu_mag[2];
    double c_gr[3], c_gr_mag, dc_gr[3][2], dc_gr_mag[2];
} src_rd;

struct grid3d;
struct gridg;

typedef struct {          /* EquationSets.2DStratified.cpp:24-30 */
    double c_eff, c_eff_0, c_eff_diff, c_eff_ddiff;
} src_2d;

struct orc_ctx {
    int eqset;
    /* atmosphere (G2S_*Spline1D.cpp globals) */
    int n;
    double *x, *T, *u, *v, *rho, *sT, *su, *sv, *srho;
    spline1d Temp, Windu, Windv, Dens;
    double x_min, x_max;           /* z_min/z_max or r_min/r_max */
    double r_earth;                /* 6370 for Global (G2S_GlobalSpline1D.cpp:35), 0 otherwise */
    double z_grnd, tweak_abs;
    /* GeoAc.Parameters*.cpp */
    double theta, phi;
    int    EqCnt, CalcAmp;
    double ds_min, ds_max, ray_limit, vert_limit, range_limit;
    /* sources */
    src_global G; src_3d S3; src_2d S2; src_rd RD;
    /* range-dependent sets: grid atmosphere + lateral limits (GeoAc.Parameters.RngDep.cpp:24-28) */
    struct grid3d* G3;
    struct gridg* GG;              /* Global.RngDep: x/y limits hold the lat/lon box (GeoAc.Parameters.Global.cpp:26-30) */
    double x_min_limit, x_max_limit, y_min_limit, y_max_limit;
    /* solution array, rows of EqCnt doubles (Interface.cpp:53-58), kept contiguous with stride 18 */
    double* sol; int64_t sol_rows;
};
#define SOLSTRIDE 18
#define ROW(ctx,k) ((ctx)->sol + (size_t)(k) * SOLSTRIDE)

orc_ctx* orc_create(int eqset){
    if(eqset != GEOAC_EQ_2D && eqset != GEOAC_EQ_3D && eqset != GEOAC_EQ_GLOBAL && eqset != GEOAC_EQ_3D_RNGDEP && eqset != GEOAC_EQ_GLOBAL_RNGDEP) return NULL;
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
    c->eqset = eqset;
    c->ds_min = 0.001; c->ds_max = 0.5;                       /* Parameters.cpp:19-20 */
    if(eqset == GEOAC_EQ_GLOBAL || eqset == GEOAC_EQ_GLOBAL_RNGDEP){ c->ray_limit = 10000.0; c->r_earth = 6370.0; }   /* Parameters.Global.cpp:23 */
    else { c->ray_limit = 5000.0; c->vert_limit = 200.0; c->range_limit = 2000.0; c->r_earth = 0.0; }  /* Parameters.cpp:23-25 */
    if(eqset == GEOAC_EQ_3D_RNGDEP){ c->vert_limit = 160.0; c->x_min_limit = -500.0; c->x_max_limit = 500.0; c->y_min_limit = -500.0; c->y_max_limit = 500.0; }  /* Parameters.RngDep.cpp:24-28 */
    c->tweak_abs = 0.3; c->z_grnd = 0.0;
    return c;
}

static void free_atmo(orc_ctx* c){
    free(c->x); free(c->T); free(c->u); free(c->v); free(c->rho);
    free(c->sT); free(c->su); free(c->sv); free(c->srho);
    c->x = c->T = c->u = c->v = c->rho = c->sT = c->su = c->sv = c->srho = NULL; c->n = 0;
}
static void g3_free(struct grid3d* G);
static void gg_free(struct gridg* G);
void orc_destroy(orc_ctx* c){ if(!c) return; free_atmo(c); g3_free(c->G3); gg_free(c->GG); free(c->sol); free(c); }

static void alloc_atmo(orc_ctx* c, int n){
    free_atmo(c); c->n = n;
    size_t b = sizeof(double) * (size_t)n;
    c->x = malloc(b); c->T = malloc(b); c->u = malloc(b); c->v = malloc(b); c->rho = malloc(b);
    c->sT = malloc(b); c->su = malloc(b); c->sv = malloc(b); c->srho = malloc(b);
}

/* Spline_Single_G2S after the columns are in place: G2S_Spline1D.cpp:293-309, G2S_GlobalSpline1D.cpp:305-320;
 * GeoAc_SetPropRegion: G2S_Spline1D.cpp:22-28 / G2S_GlobalSpline1D.cpp:22-30 */
static void finish_load(orc_ctx* c){
    int n = c->n;
    spline1d* S[4] = { &c->Temp, &c->Windu, &c->Dens, &c->Windv };
    double* F[4] = { c->T, c->u, c->rho, c->v }; double* SL[4] = { c->sT, c->su, c->srho, c->sv };
    for(int i = 0; i < 4; i++){ S[i]->length = n; S[i]->accel = 0; S[i]->x_vals = c->x; S[i]->f_vals = F[i]; S[i]->slopes = SL[i]; }
    c->x_min = c->x[0]; c->x_max = c->x[n-1];
    c->vert_limit = c->x_max;
    c->range_limit = (c->eqset == GEOAC_EQ_GLOBAL) ? 1500.0 : 10000.0;
    for(int i = 0; i < 4; i++) set_slopes(S[i]);
}

/* taper + unit conversion of Load_G2S: G2S_Spline1D.cpp:120-124, G2S_GlobalSpline1D.cpp:127-130.
 * z_grnd is still 0 here: the mains load the profile before parsing z_grnd= (Q9). */
static void taper_row(orc_ctx* c, int i){
    double zg = 0.0;
    if(c->eqset == GEOAC_EQ_GLOBAL){
        c->x[i] += c->r_earth;
        c->u[i] *= (2.0 / (1.0 + exp(-(c->x[i] - c->r_earth - zg)/0.2)) - 1.0) / 1000.0;
        c->v[i] *= (2.0 / (1.0 + exp(-(c->x[i] - c->r_earth - zg)/0.2)) - 1.0) / 1000.0;
    } else {
        c->u[i] *= (2.0 / (1.0 + exp(-(c->x[i] - zg)/0.2)) - 1.0) / 1000.0;
        c->v[i] *= (2.0 / (1.0 + exp(-(c->x[i] - zg)/0.2)) - 1.0) / 1000.0;
    }
}

int orc_load_arrays(orc_ctx* c, int n, const double* z, const double* T, const double* u, const double* v, const double* rho){
    if(n < 3) return -1;
    alloc_atmo(c, n);
    for(int i = 0; i < n; i++){
        c->x[i] = z[i]; c->T[i] = T[i]; c->u[i] = u[i]; c->v[i] = v[i]; c->rho[i] = rho[i];
        taper_row(c, i);
    }
    finish_load(c);
    return n;
}

/* file_length (row count = number of '\n', G2S_Spline1D.cpp:55-71) + Load_G2S (:109-142) */
int orc_load(orc_ctx* c, const char* met_file, const char* format){
    FILE* fp = fopen(met_file, "r");
    if(!fp) return -1;
    int rows = 0, ch;
    while((ch = fgetc(fp)) != EOF) if(ch == '\n') rows++;
    rewind(fp);
    int fmt;
    if(strncmp(format, "zTuvdp", 6) == 0) fmt = 0;
    else if(strncmp(format, "zuvwTdp", 7) == 0) fmt = 1;
    else { fclose(fp); return -2; }
    if(rows < 3){ fclose(fp); return -3; }
    alloc_atmo(c, rows);
    for(int i = 0; i < rows; i++){
        double t[7]; int nt = fmt ? 7 : 6;
        for(int j = 0; j < nt; j++) if(fscanf(fp, "%lf", &t[j]) != 1) t[j] = 0.0;
        if(fmt == 0){ c->x[i] = t[0]; c->T[i] = t[1]; c->u[i] = t[2]; c->v[i] = t[3]; c->rho[i] = t[4]; }
        else        { c->x[i] = t[0]; c->u[i] = t[1]; c->v[i] = t[2]; c->T[i] = t[4]; c->rho[i] = t[5]; }
        taper_row(c, i);
    }
    fclose(fp);
    finish_load(c);
    return rows;
}

void orc_limits(orc_ctx* c, double* vl, double* rl){ *vl = c->vert_limit; *rl = c->range_limit; }
/* GeoAc_ray_limit (GeoAc.Parameters{,.Global}.cpp:23): the reference's step bound is ray_limit * int(1 / (10 ds_min)), GeoAc.Solver.cpp:14 */
void orc_set_ray_limit(orc_ctx* c, double ray_limit){ c->ray_limit = ray_limit; }

/* ------------------------------------------------------------------------------------------ */
/* Atmo_State.h scalar API on the spline abscissa: G2S_Spline1D.cpp:321-416, Global twin :332-428 */
/* ------------------------------------------------------------------------------------------ */
static double clampx(const orc_ctx* c, double x){ double e = DMIN(x, c->x_max); e = DMAX(e, c->x_min); return e; }

static double atm_rho(orc_ctx* c, double x){ return eval_f(clampx(c, x), &c->Dens); }
static double atm_c(orc_ctx* c, double x){ return sqrt(gamR * eval_f(clampx(c, x), &c->Temp)); }
static double atm_c_diff(orc_ctx* c, double x){
    double e = clampx(c, x);
    return gamR / (2.0 * atm_c(c, x)) * eval_df(e, &c->Temp);
}
static double atm_c_ddiff(orc_ctx* c, double x){
    double e = clampx(c, x);
    double SndSpd = atm_c(c, x);
    return gamR / (2.0 * SndSpd) * eval_ddf(e, &c->Temp)
         - pow(gamR,2)/(4.0 * pow(SndSpd,3)) * pow(eval_df(e, &c->Temp),2);
}
static double atm_u(orc_ctx* c, double x){ return eval_f(clampx(c, x), &c->Windu); }
static double atm_u_diff(orc_ctx* c, double x){ return eval_df(clampx(c, x), &c->Windu); }
static double atm_u_ddiff(orc_ctx* c, double x){ return eval_ddf(clampx(c, x), &c->Windu); }
static double atm_v(orc_ctx* c, double x){ return eval_f(clampx(c, x), &c->Windv); }
static double atm_v_diff(orc_ctx* c, double x){ return eval_df(clampx(c, x), &c->Windv); }
static double atm_v_ddiff(orc_ctx* c, double x){ return eval_ddf(clampx(c, x), &c->Windv); }
/* w, w_diff, w_ddiff are identically 0 (G2S_Spline1D.cpp:414-416) */

/* SuthBass_Alpha: Atmo_State.Absorption.cpp:14-143 / Atmo_State.Absorption.Global.cpp:12-141.
 * `x` is the spline abscissa (z or geocentric r); zr = altitude above sea level used by the gas-fraction fits.
 * Global quirk kept: the reference temperature/pressure are taken at abscissa z_grnd (a km altitude passed
 * as a radius, so it clamps to the lowest node), Absorption.Global.cpp:31-32. */
static double suthbass_core(orc_ctx* c, double zr, double c_grnd, double rho_grnd, double c_pt, double rho_pt, double freq){
    double T_o, P_o, S, X[7], X_ON, Z_rot[2], Z_rot_;
    double sigma, nn, chi, cchi, mu, nu, mu_o;
    double a_cl, a_rot, a_diff, a_vib;
    double T_z, P_z, c_snd_z;
    double A1, A2, B, C, D, E, F, G, H, I, J, K, L, ZZ, hu;
    double f_vib[4], a_vib_c[4], Cp_R[4], Cv_R[4], Theta[4], C_R, A_max, Tr;

    mu_o  = 18.192E-6;
    T_o   = pow(c_grnd*1000.0,2)/(Rgas*gam);
    P_o   = rho_grnd*pow(c_grnd*1000.0,2)/gam*1000.0;
    S     = 117.0;

    Cv_R[0] = 5.0/2.0; Cv_R[1] = 5.0/2.0; Cv_R[2] = 3.0; Cv_R[3] = 3.0;
    Cp_R[0] = 7.0/2.0; Cp_R[1] = 7.0/2.0; Cp_R[2] = 4.0; Cp_R[3] = 4.0;
    Theta[0]= 2239.1;  Theta[1]= 3352.0;  Theta[2]= 915.0; Theta[3]= 1037.0;

    T_z     = pow(c_pt*1000.0,2)/(Rgas*gam);
    P_z     = rho_pt*pow(c_pt*1000.0,2)/gam * 1000.0;
    c_snd_z = c_pt;

    mu      = mu_o*sqrt(T_z/T_o)*((1.0+S/T_o)/(1.0+S/T_z));
    nu      = (8.0*Pi*freq*mu)/(3.0*P_z);

    if (zr > 90.)  X[0] = pow(10.0,49.296-(1.5524*zr)+(1.8714E-2*pow(zr,2))-(1.1069E-4*pow(zr,3))+(3.199E-7*pow(zr,4))-(3.6211E-10*pow(zr,5)));
    else           X[0] = pow(10.0,-0.67887);
    if (zr > 76.)  X[1] = pow(10.0,(1.3972E-1)-(5.6269E-3*zr)+(3.9407E-5*pow(zr,2))-(1.0737E-7*pow(zr,3)));
    else           X[1] = pow(10.0,-0.10744);
    X[2]  = pow(10,-3.3979);
    if (zr > 80. ) X[3] = pow(10.0,-4.234-(3.0975E-2*zr));
    else           X[3] = pow(10.0,-19.027+(1.3093*zr)-(4.6496E-2*pow(zr,2))+(7.8543E-4*pow(zr,3))-(6.5169E-6*pow(zr,4))+(2.1343E-8*pow(zr,5)));
    if (zr > 95. ) X[4] = pow(10.0,-3.2456+(4.6642E-2*zr)-(2.6894E-4*pow(zr,2))+(5.264E-7*pow(zr,3)));
    else           X[4] = pow(10.0,-11.195+(1.5408E-1*zr)-(1.4348E-3*pow(zr,2))+(1.0166E-5*pow(zr,3)));
    X[5]  = pow(10.0,-53.746+(1.5439*zr)-(1.8824E-2*pow(zr,2))+(1.1587E-4*pow(zr,3))-(3.5399E-7*pow(zr,4))+(4.2609E-10*pow(zr,5)));
    if (zr > 30. ) X[6] = pow(10.0,-4.2563+(7.6245E-2*zr)-(2.1824E-3*pow(zr,2))-(2.3010E-6*pow(zr,3))+(2.4265E-7*pow(zr,4))-(1.2500E-09*pow(zr,5)));
    else { if (zr > 100.) X[6] = pow(10.0,-0.62534-(8.3665E-2*zr));
           else           X[6] = pow(10.0,-1.7491+(4.4986E-2*zr)-(6.8549E-2*pow(zr,2))+(5.4639E-3*pow(zr,3))-(1.5539E-4*pow(zr,4))+(1.5063E-06*pow(zr,5))); }

    X_ON = (X[0] + X[1])/0.9903;

    Z_rot[0] = 54.1*exp(-17.3*(pow(T_z,-1.0/3.0)));
    Z_rot[1] = 63.3*exp(-16.7*(pow(T_z,-1.0/3.0)));
    Z_rot_   = 1.0/((X[1]/Z_rot[1])+(X[0]/Z_rot[0]));

    sigma = 5.0/sqrt(21.0);
    nn = (4.0/5.0)*sqrt(3.0/7.0)*Z_rot_;
    chi=3.0*nn*nu/4.0;
    cchi=2.36*chi;

    a_cl    = (2.0*Pi*freq/c_snd_z)*sqrt(0.5*(sqrt(1.0+pow(nu,2))-1.0)*(1.0+pow(cchi,2))/((1.0+pow(nu,2))*(1.0+pow(sigma*cchi,2))));
    a_rot   = (2.0*Pi*freq/c_snd_z)*X_ON*((pow(sigma,2)-1.0)*chi/(2*sigma))*sqrt(0.5*(sqrt(1.0+pow(nu,2))+1.0)/((1.0+pow(nu,2))*(1.0+pow(cchi,2))));
    a_diff  = 0.003*a_cl;

    Tr = pow(T_z/T_o,-1.0/3.0)-1.0;
    A1 = (X[0]+X[1])*24.0*exp(-9.16*Tr);
    A2 = (X[4]+X[5])*2400.0;
    B  = 40400.0*exp(10.0*Tr);
    C  = 0.02*exp(-11.2*Tr);
    D  = 0.391*exp(8.41*Tr);
    E  = 9.0*exp(-19.9*Tr);
    F  = 60000.0;
    G  = 28000.0*exp(-4.17*Tr);
    H  = 22000.0*exp(-7.68*Tr);
    I  = 15100.0*exp(-10.4*Tr);
    J  = 11500.0*exp(-9.17*Tr);
    K  = (8.48E08)*exp(9.17*Tr);
    L  = exp(-7.72*Tr);
    ZZ = H*X[2]+I*(X[0]+0.5*X[4])+J*(X[1]+0.5*X[5])+K*(X[6]+X[3]);
    hu = 100.0*(X[3]+X[6]);
    f_vib[0] = (P_z/P_o)*(mu_o/mu)*(A1+A2+B*hu*(C+hu)*(D+hu));
    f_vib[1] = (P_z/P_o)*(mu_o/mu)*(E+F*X[3]+G*X[6]);
    f_vib[2] = (P_z/P_o)*(mu_o/mu)*ZZ;
    f_vib[3] = (P_z/P_o)*(mu_o/mu)*(1.2E5)*L;

    a_vib = 0.0;
    for (int m=0; m<4; m++){
        C_R          = ((pow(Theta[m]/T_z,2))*exp(-Theta[m]/T_z))/(pow(1-exp(-Theta[m]/T_z),2));
        A_max        = (X[m]*(Pi/2)*C_R)/(Cp_R[m]*(Cv_R[m]+C_R));
        a_vib_c[m]   = (A_max/c_snd_z)*((2*(pow(freq,2))/f_vib[m])/(1+pow(freq/f_vib[m],2)));
        a_vib        += a_vib_c[m];
    }
    return (a_cl + a_rot + a_diff + a_vib) * c->tweak_abs * 8.685889;
}

/* 1-D atmosphere wrapper: the medium values SuthBass_Alpha looks up, in its call order (ground state first) */
static double suthbass_alpha(orc_ctx* c, double x, double freq){
    double c_g = atm_c(c, c->z_grnd), rho_g = atm_rho(c, c->z_grnd);
    double c_z = atm_c(c, x), rho_z = atm_rho(c, x);
    return suthbass_core(c, x - c->r_earth, c_g, rho_g, c_z, rho_z, freq);
}

/* ------------------------------------------------------------------------------------------ */
/* Global equation set: GeoAc.EquationSets.Global.cpp                                           */
/* ------------------------------------------------------------------------------------------ */
/* GeoAc_SetInitialConditions: Global.cpp:76-136.  r0 = altitude, theta0/phi0 = lat/lon [rad] */
static void g_set_ic(orc_ctx* c, double r0, double theta0, double phi0){
    src_global* S = &c->G; double* y = ROW(c, 0);
    double re = c->r_earth;
    S->src_loc[0] = r0 + re; S->src_loc[1] = theta0; S->src_loc[2] = phi0;
    S->c0 = atm_c(c, r0 + re);
    double MachComps[3] = { 0.0/S->c0, atm_v(c, r0 + re)/S->c0, atm_u(c, r0 + re)/S->c0 };
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

/* GeoAc_Set_ds: Global.cpp:210-217 (3DStratified.cpp:191-198, 2DStratified.cpp:123-130 with z) */
static double set_ds(const orc_ctx* c, double height_above_ground){
    double result = 0.05 - 0.049 * exp(-(height_above_ground)/0.75);
    result = DMIN(result, c->ds_max);
    result = DMAX(result, c->ds_min);
    return result;
}

/* GeoAc_UpdateSources: Global.cpp:222-370.  Stratified atmosphere: every d/dtheta, d/dphi and every
 * w-term of the Atmo_State API returns 0.0 (G2S_GlobalSpline1D.cpp:351-428); they are kept as explicit
 * zeros so signed-zero / operand-order behaviour is unchanged. */
static void g_update_sources(orc_ctx* c, const double* y){
    src_global* S = &c->G;
    double r = y[0], theta = y[1];
    double nu[3] = { y[3], y[4], y[5] };
    double dcn[3] = {0,0,0}, dwn[3] = {0,0,0}, dvn[3] = {0,0,0}, dun[3] = {0,0,0};

    S->c = atm_c(c, r); S->w = 0.0; S->v = atm_v(c, r); S->u = atm_u(c, r);
    dcn[0] = atm_c_diff(c, r); dvn[0] = atm_v_diff(c, r); dun[0] = atm_u_diff(c, r);
    for(int n = 0; n < 3; n++){ S->dc[n] = dcn[n]; S->dw[n] = dwn[n]; S->dv[n] = dvn[n]; S->du[n] = dun[n]; }

    S->nu_mag = sqrt(nu[0]*nu[0] + nu[1]*nu[1] + nu[2]*nu[2]);
    S->c_gr[0] = S->c*nu[0]/S->nu_mag + S->w;
    S->c_gr[1] = S->c*nu[1]/S->nu_mag + S->v;
    S->c_gr[2] = S->c*nu[2]/S->nu_mag + S->u;
    S->c_gr_mag = sqrt(pow(S->c_gr[0],2) + pow(S->c_gr[1],2) + pow(S->c_gr[2],2));

    S->GeoCoeff[0] = 1.0;
    S->GeoCoeff[1] = 1.0/r;
    S->GeoCoeff[2] = 1.0/(r*cos(theta));

    S->GeoTerms[0] = 0.0;
    S->GeoTerms[1] = (nu[0]*S->v - nu[1]*S->w);
    S->GeoTerms[2] = (nu[0]*S->u - nu[2]*S->w)*cos(theta) + (nu[1]*S->u - nu[2]*S->v)*sin(theta);
    S->GeoTerms[0] += 1.0/r * (nu[1]*S->c_gr[1] + nu[2]*S->c_gr[2]);
    S->GeoTerms[1] += -nu[0]*S->c_gr[1] + nu[2]*S->c_gr[2]*tan(theta);
    S->GeoTerms[2] += -S->c_gr[2]*(nu[0]*cos(theta) + nu[1]*sin(theta));

    if(!c->CalcAmp) return;

    /* a = 0: d/d(launch inclination) "lt";  a = 1: d/d(launch azimuth) "lp" */
    double Rl[2][3] = { { y[6],  y[7],  y[8]  }, { y[12], y[13], y[14] } };
    double ml[2][3] = { { y[9],  y[10], y[11] }, { y[15], y[16], y[17] } };
    double ddcn[3][3] = {{0}}, ddvn[3][3] = {{0}}, ddun[3][3] = {{0}};      /* [m][n] second derivatives; only rr != 0 */
    ddcn[0][0] = atm_c_ddiff(c, r); ddvn[0][0] = atm_v_ddiff(c, r); ddun[0][0] = atm_u_ddiff(c, r);

    for(int a = 0; a < 2; a++){
        S->dc[3+a] = 0.0; S->dw[3+a] = 0.0; S->dv[3+a] = 0.0; S->du[3+a] = 0.0;
        for(int m = 0; m < 3; m++){ S->ddc[m][a] = 0.0; S->ddw[m][a] = 0.0; S->ddv[m][a] = 0.0; S->ddu[m][a] = 0.0; }
    }
    for(int n = 0; n < 3; n++){
        for(int a = 0; a < 2; a++){
            S->dc[3+a] += Rl[a][n]*dcn[n];
            S->dw[3+a] += Rl[a][n]*dwn[n];
            S->dv[3+a] += Rl[a][n]*dvn[n];
            S->du[3+a] += Rl[a][n]*dun[n];
        }
        for(int m = 0; m < 3; m++){
            for(int a = 0; a < 2; a++){
                S->ddc[m][a] += Rl[a][n]*ddcn[m][n];
                S->ddw[m][a] += Rl[a][n]*0.0;
                S->ddv[m][a] += Rl[a][n]*ddvn[m][n];
                S->ddu[m][a] += Rl[a][n]*ddun[m][n];
            }
        }
    }
    double wind_d[3][2] = { { S->dw[3], S->dw[4] }, { S->dv[3], S->dv[4] }, { S->du[3], S->du[4] } };
    for(int a = 0; a < 2; a++){
        const double* R_l = Rl[a]; const double* mu_l = ml[a];
        S->dnu_mag[a] = (nu[0]*mu_l[0] + nu[1]*mu_l[1] + nu[2]*mu_l[2])/S->nu_mag;
        for(int i = 0; i < 3; i++)
            S->dc_gr[i][a] = nu[i]/S->nu_mag*S->dc[3+a] + S->c*mu_l[i]/S->nu_mag - S->c*nu[i]/pow(S->nu_mag,2) * S->dnu_mag[a] + wind_d[i][a];
        S->dc_gr_mag[a] = (S->c_gr[0]*S->dc_gr[0][a] + S->c_gr[1]*S->dc_gr[1][a] + S->c_gr[2]*S->dc_gr[2][a])/S->c_gr_mag;

        S->d_GeoCoeff[0][a] = 0.0;
        S->d_GeoCoeff[1][a] = -R_l[0]/(pow(r,2));
        S->d_GeoCoeff[2][a] = -R_l[0]/(pow(r,2)*cos(theta)) + sin(theta)/(r*pow(cos(theta),2))*R_l[1];

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

/* GeoAc_EvalSrcEq: Global.cpp:374-442 */
static double g_eval_src_eq(const orc_ctx* c, const double* y, int q){
    const src_global* S = &c->G;
    double nu[3] = { y[3], y[4], y[5] };
    if(q < 3) return S->GeoCoeff[q]*S->c_gr[q]/S->c_gr_mag;
    if(q < 6){
        int i = q - 3;
        return -S->GeoCoeff[i]/S->c_gr_mag*(S->nu_mag*S->dc[i] + nu[0]*S->dw[i] + nu[1]*S->dv[i] + nu[2]*S->du[i] + S->GeoTerms[i]);
    }
    int a = (q >= 12) ? 1 : 0;
    int i = (q - 6) % 3;
    if((q - 6) % 6 < 3){
        return S->d_GeoCoeff[i][a]*S->c_gr[i]/S->c_gr_mag
             + S->GeoCoeff[i]*S->dc_gr[i][a]/S->c_gr_mag
             - S->GeoCoeff[i]*S->c_gr[i]/pow(S->c_gr_mag,2) * S->dc_gr_mag[a];
    }
    const double* mu = a ? (y + 15) : (y + 9);
    return -S->d_GeoCoeff[i][a]/S->c_gr_mag*(S->nu_mag*S->dc[i] + nu[0]*S->dw[i] + nu[1]*S->dv[i] + nu[2]*S->du[i] + S->GeoTerms[i])
         + S->GeoCoeff[i]/pow(S->c_gr_mag,2) * S->dc_gr_mag[a]*(S->nu_mag*S->dc[i] + nu[0]*S->dw[i] + nu[1]*S->dv[i] + nu[2]*S->du[i])
         - S->GeoCoeff[i]/S->c_gr_mag*(S->dnu_mag[a]*S->dc[i] + S->nu_mag*S->ddc[i][a]
              + mu[0]*S->dw[i] + mu[1]*S->dv[i] + mu[2]*S->du[i]
              + nu[0]*S->ddw[i][a] + nu[1]*S->ddv[i][a] + nu[2]*S->ddu[i][a] + S->d_GeoTerms[i][a]);
}

/* GeoAc_BreakCheck / GeoAc_GroundCheck: Global.cpp:500-522 */
static int g_break_check(const orc_ctx* c, int k){
    const double* y = ROW(c, k); const src_global* S = &c->G;
    double GC_Dist1 = pow(sin((y[1] - S->src_loc[1])/2.0),2);
    double GC_Dist2 = cos(S->src_loc[1]) * cos(y[1]) * pow(sin((y[2] - S->src_loc[2])/2.0),2);
    double range = 2.0 * c->r_earth * asin(sqrt(GC_Dist1 + GC_Dist2));
    int check = 0;
    if(y[0] > c->vert_limit) check = 1;
    if(range > c->range_limit) check = 1;
    return check;
}
static int g_ground_check(const orc_ctx* c, int k){ return ROW(c, k)[0] < (c->r_earth + c->z_grnd); }

/* one segment of GeoAc_TravelTime / GeoAc_TravelTimeSegment: Global.cpp:527-589 */
static double g_tt_seg(orc_ctx* c, int n){
    const double* a = ROW(c, n); const double* b = ROW(c, n+1);
    double dr = b[0] - a[0], dt = b[1] - a[1], dp = b[2] - a[2];
    double r = a[0] + dr/2.0, t = a[1] + dt/2.0;
    double ds = sqrt(pow(dr,2) + pow(r*dt,2) + pow(r*cos(t)*dp,2));
    double nu[3];
    nu[0] = a[3] + (b[3] - a[3])/2.0;
    nu[1] = a[4] + (b[4] - a[4])/2.0;
    nu[2] = a[5] + (b[5] - a[5])/2.0;
    double nu_mag = sqrt(nu[0]*nu[0] + nu[1]*nu[1] + nu[2]*nu[2]);
    double cc = atm_c(c, r), vv = atm_v(c, r), uu = atm_u(c, r);
    double c_prop[3] = { cc*nu[0]/nu_mag + 0.0, cc*nu[1]/nu_mag + vv, cc*nu[2]/nu_mag + uu };
    double c_prop_mag = sqrt(pow(c_prop[0],2) + pow(c_prop[1],2) + pow(c_prop[2],2));
    return ds/c_prop_mag;
}
/* one segment of GeoAc_SB_Atten(+Segment): Global.cpp:634-670 (sin(t) in ds: Q3) */
static double g_att_seg(orc_ctx* c, int n, double freq){
    const double* a = ROW(c, n); const double* b = ROW(c, n+1);
    double dr = b[0] - a[0], dt = b[1] - a[1], dp = b[2] - a[2];
    double r = a[0] + dr/2.0, t = a[1] + dt/2.0;
    double ds = sqrt(pow(dr,2) + pow(r*dt,2) + pow(r*sin(t)*dp,2));
    return suthbass_alpha(c, r, freq)*ds;
}

/* GeoAc_Jacobian: Global.cpp:594-607 (1/(r sin) for dp_ds: Q3) */
static double g_jacobian(orc_ctx* c, int k){
    const double* y = ROW(c, k);
    double r = y[0], theta = y[1];
    double nu[3] = { y[3], y[4], y[5] };
    double nu_mag = sqrt(nu[0]*nu[0] + nu[1]*nu[1] + nu[2]*nu[2]);
    double cc = atm_c(c, r), vv = atm_v(c, r), uu = atm_u(c, r);
    double c_prop[3] = { cc*nu[0]/nu_mag + 0.0, cc*nu[1]/nu_mag + vv, cc*nu[2]/nu_mag + uu };
    double c_prop_mag = sqrt(pow(c_prop[0],2) + pow(c_prop[1],2) + pow(c_prop[2],2));
    double dr_ds = c_prop[0]/c_prop_mag, dt_ds = 1.0/r*c_prop[1]/c_prop_mag, dp_ds = 1.0/(r*sin(theta))*c_prop[2]/c_prop_mag;
    double dr_dlt = y[6],  dt_dlt = y[7],  dp_dlt = y[8];
    double dr_dlp = y[12], dt_dlp = y[13], dp_dlp = y[14];
    return pow(r,2)*cos(theta)*(dr_ds*(dt_dlt*dp_dlp - dt_dlp*dp_dlt) - dr_dlt*(dt_ds*dp_dlp - dp_ds*dt_dlp) + dr_dlp*(dt_ds*dp_dlt - dp_ds*dt_dlt));
}

/* GeoAc_Amplitude: Global.cpp:610-629 (c_prop0[1..2] divided by the ARRIVAL nu_mag: Q4) */
static double g_amplitude(orc_ctx* c, int k){
    const src_global* S = &c->G; const double* y = ROW(c, k);
    double r0 = S->src_loc[0];
    double r = y[0];
    double nu[3] = { y[3], y[4], y[5] };
    double th = c->theta, ph = c->phi;
    double nu0[3] = { sin(th), cos(th)*sin(ph), cos(th)*cos(ph) };
    double cc = atm_c(c, r), vv = atm_v(c, r), uu = atm_u(c, r);
    double nu_mag = (S->c0 - nu[0]*0.0 - nu[1]*vv - nu[2]*uu)/cc;
    double nu_mag0 = S->nu0;
    double c_prop[3]  = { cc*nu[0]/nu_mag + 0.0, cc*nu[1]/nu_mag + vv, cc*nu[2]/nu_mag + uu };
    double c_prop0[3] = { S->c0*nu0[0]/nu_mag0 + 0.0, S->c0*nu0[1]/nu_mag + atm_v(c, r0), S->c0*nu0[2]/nu_mag + atm_u(c, r0) };
    double c_prop_mag  = sqrt(pow(c_prop[0],2) +  pow(c_prop[1],2) +  pow(c_prop[2],2));
    double c_prop_mag0 = sqrt(pow(c_prop0[0],2) + pow(c_prop0[1],2) + pow(c_prop0[2],2));
    double D = g_jacobian(c, k);
    double Amp_Num = atm_rho(c, r) * nu_mag * pow(atm_c(c, r),3) * c_prop_mag0 * cos(th);
    double Amp_Den = atm_rho(c, r0)* nu_mag0* pow(atm_c(c, r0),3)* c_prop_mag  * D;
    return 1.0/(4.0*Pi)*sqrt(fabs(Amp_Num/Amp_Den));
}

/* GeoAc_ApproximateIntercept + GeoAc_SetReflectionConditions: Global.cpp:140-205.
 * Q1: the stray ';' at Global.cpp:146 drops the quadratic term - linear intercept only. */
static void g_reflect(orc_ctx* c, int k){
    const src_global* S = &c->G;
    double prev[18];
    const double* yk = ROW(c, k); const double* ykm = ROW(c, k-1);
    double rg = c->r_earth + c->z_grnd;
    double dr_k = yk[0] - ykm[0];
    double dr_grnd = ykm[0] - rg;
    for(int i = 0; i < c->EqCnt; i++) prev[i] = ykm[i] + (ykm[i] - yk[i])/dr_k*dr_grnd;

    double c_ref = atm_c(c, prev[0]);
    double dnu_r_ds = - 1.0/c_ref * (S->c0/c_ref * atm_c_diff(c, prev[0])
                                     + prev[3] * 0.0
                                     + prev[4] * atm_v_diff(c, prev[0])
                                     + prev[5] * atm_u_diff(c, prev[0])
                                     + c_ref/prev[0] * (pow(prev[4],2) + pow(prev[5],2)));
    double* y0 = ROW(c, 0);
    for(int i = 0; i < c->EqCnt; i++){
        if(i == 0) y0[i] = rg;
        else if(i == 3 || i == 6 || i == 12) y0[i] = -prev[i];
        else if(i == 9 || i == 15) y0[i] = -prev[i] + 2.0*dnu_r_ds * prev[i-3]/(c_ref / S->c0 * prev[3]);
        else y0[i] = prev[i];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* 3-D stratified Cartesian set: GeoAc.EquationSets.3DStratified.cpp                            */
/* ------------------------------------------------------------------------------------------ */
/* GeoAc_SetInitialConditions: 3DStratified.cpp:69-131 */
static void s3_set_ic(orc_ctx* c, double x0, double y0, double z0){
    src_3d* S = &c->S3; double* y = ROW(c, 0);
    S->src_loc[0] = x0; S->src_loc[1] = y0; S->src_loc[2] = z0;
    S->c0 = atm_c(c, z0);
    double M_Comps[3] = { atm_u(c, z0)/S->c0, atm_v(c, z0)/S->c0, 0.0/S->c0 };
    double th = c->theta, ph = c->phi;
    double nu0[3]    = { cos(th)*cos(ph),  cos(th)*sin(ph), sin(th) };
    double mu0_th[3] = {-sin(th)*cos(ph), -sin(th)*sin(ph), cos(th) };
    double mu0_ph[3] = {-cos(th)*sin(ph),  cos(th)*cos(ph), 0.0 };
    double M = 1.0 + (nu0[0]*M_Comps[0] + nu0[1]*M_Comps[1] + nu0[2]*M_Comps[2]);
    double dM_th = mu0_th[0]*M_Comps[0] + mu0_th[1]*M_Comps[1] + mu0_th[2]*M_Comps[2];
    double dM_ph = mu0_ph[0]*M_Comps[0] + mu0_ph[1]*M_Comps[1] + mu0_ph[2]*M_Comps[2];
    S->nu0_xy[0] = nu0[0]/M; S->nu0_xy[1] = nu0[1]/M;
    S->mu0_xy[0][0] = mu0_th[0]/M - nu0[0]/pow(M,2.0)*dM_th;
    S->mu0_xy[1][0] = mu0_th[1]/M - nu0[1]/pow(M,2.0)*dM_th;
    S->mu0_xy[0][1] = mu0_ph[0]/M - nu0[0]/pow(M,2.0)*dM_ph;
    S->mu0_xy[1][1] = mu0_ph[1]/M - nu0[1]/pow(M,2.0)*dM_ph;
    for(int i = 0; i < c->EqCnt; i++){
        switch(i){
            case 0: y[i] = x0; break;
            case 1: y[i] = y0; break;
            case 2: y[i] = z0; break;
            case 3: y[i] = nu0[2]/M; break;
            case 7: y[i] = mu0_th[2]/M - nu0[2]/pow(M,2.0)*dM_th; break;
            case 11: y[i] = mu0_ph[2]/M - nu0[2]/pow(M,2.0)*dM_ph; break;
            default: y[i] = 0.0;
        }
    }
}

/* GeoAc_UpdateSources: 3DStratified.cpp:203-246 */
static void s3_update_sources(orc_ctx* c, const double* y){
    src_3d* S = &c->S3;
    double z = y[2];
    double nu[3] = { S->nu0_xy[0], S->nu0_xy[1], y[3] };
    S->c = atm_c(c, z);  S->dc = atm_c_diff(c, z);
    S->u = atm_u(c, z);  S->du = atm_u_diff(c, z);
    S->v = atm_v(c, z);  S->dv = atm_v_diff(c, z);
    S->w = 0.0;          S->dw = 0.0;
    S->nu_mag = S->c0/S->c * (1.0 - (nu[0]*S->u + nu[1]*S->v + nu[2]*S->w)/S->c0);
    S->c_prop[0] = S->c*nu[0]/S->nu_mag + S->u;
    S->c_prop[1] = S->c*nu[1]/S->nu_mag + S->v;
    S->c_prop[2] = S->c*nu[2]/S->nu_mag + S->w;
    S->c_prop_mag = sqrt(pow(S->c_prop[0],2) + pow(S->c_prop[1],2) + pow(S->c_prop[2],2));
    if(!c->CalcAmp) return;
    S->ddc = atm_c_ddiff(c, z); S->ddu = atm_u_ddiff(c, z); S->ddv = atm_v_ddiff(c, z); S->ddw = 0.0;
    double mu[2][3] = { { S->mu0_xy[0][0], S->mu0_xy[1][0], y[7] }, { S->mu0_xy[0][1], S->mu0_xy[1][1], y[11] } };
    double Za[2] = { y[6], y[10] };
    double dwinds[3] = { atm_u_diff(c, z), atm_v_diff(c, z), 0.0 };
    for(int a = 0; a < 2; a++)
        S->dnu_mag[a] = (nu[0]*mu[a][0] + nu[1]*mu[a][1] + nu[2]*mu[a][2])/S->nu_mag;
    for(int n = 0; n < 3; n++){
        for(int a = 0; a < 2; a++)
            S->dc_prop[n][a] = nu[n]/S->nu_mag*S->dc*Za[a] + S->c*mu[a][n]/S->nu_mag
                             - S->c*nu[n]/pow(S->nu_mag,2)*S->dnu_mag[a] + dwinds[n]*Za[a];
    }
    for(int a = 0; a < 2; a++)
        S->dc_prop_mag[a] = (S->c_prop[0]*S->dc_prop[0][a] + S->c_prop[1]*S->dc_prop[1][a] + S->c_prop[2]*S->dc_prop[2][a])/S->c_prop_mag;
}

/* GeoAc_EvalSrcEq: 3DStratified.cpp:251-310 */
static double s3_eval_src_eq(const orc_ctx* c, const double* y, int q){
    const src_3d* S = &c->S3;
    double cp_mag = S->c_prop_mag;
    double nu[3] = { S->nu0_xy[0], S->nu0_xy[1], y[3] };
    if(q < 3) return S->c_prop[q]/cp_mag;
    if(q == 3) return -1.0/cp_mag*(S->nu_mag*S->dc + (nu[0]*S->du + nu[1]*S->dv + nu[2]*S->dw));
    int a = (q >= 8) ? 1 : 0;
    int i = q - 4 - 4*a;
    if(i < 3) return S->dc_prop[i][a]/cp_mag - S->c_prop[i]/pow(cp_mag,2)*S->dc_prop_mag[a];
    double mu[3] = { S->mu0_xy[0][a], S->mu0_xy[1][a], y[7 + 4*a] };
    double Za = y[6 + 4*a];
    return 1.0/pow(cp_mag,2)*(S->nu_mag*S->dc + (nu[0]*S->du + nu[1]*S->dv + nu[2]*S->dw))*S->dc_prop_mag[a]
         - 1.0/cp_mag*(S->dnu_mag[a]*S->dc + (mu[0]*S->du + mu[1]*S->dv + mu[2]*S->dw
                        + (S->nu_mag*S->ddc + nu[0]*S->ddu + nu[1]*S->ddv + nu[2]*S->ddw)*Za));
}

/* GeoAc_BreakCheck / GroundCheck: 3DStratified.cpp:327-343 */
static int s3_break_check(const orc_ctx* c, int k){
    const double* y = ROW(c, k);
    double r = sqrt(pow(y[0],2) + pow(y[1],2));
    int check = 0;
    if(y[2] > c->vert_limit) check = 1;
    if(r > c->range_limit) check = 1;
    return check;
}
static int s3_ground_check(const orc_ctx* c, int k){ return ROW(c, k)[2] < c->z_grnd; }

/* GeoAc_TravelTime segment: 3DStratified.cpp:348-405 (c(0,0,0) instead of c0, w ignored: Q5) */
static double s3_tt_seg(orc_ctx* c, int n){
    const src_3d* S = &c->S3;
    const double* a = ROW(c, n); const double* b = ROW(c, n+1);
    double nu[3]; nu[0] = S->nu0_xy[0]; nu[1] = S->nu0_xy[1];
    double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
    double ds = sqrt(dx*dx + dy*dy + dz*dz);
    double z = a[2] + dz/2.0;
    nu[2] = a[3] + (b[3] - a[3])/2.0;
    double c000 = atm_c(c, 0.0);
    double cc = atm_c(c, z), uu = atm_u(c, z), vv = atm_v(c, z);
    double nu_mag = (c000 - nu[0]*uu - nu[1]*vv)/cc;
    double c_prop[3] = { cc*nu[0]/nu_mag + uu, cc*nu[1]/nu_mag + vv, cc*nu[2]/nu_mag };
    double c_prop_mag = sqrt(pow(c_prop[0],2) + pow(c_prop[1],2) + pow(c_prop[2],2));
    return ds/c_prop_mag;
}
/* GeoAc_SB_Atten segment: 3DStratified.cpp:456-490 */
static double s3_att_seg(orc_ctx* c, int n, double freq){
    const double* a = ROW(c, n); const double* b = ROW(c, n+1);
    double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
    double ds = sqrt(dx*dx + dy*dy + dz*dz);
    double z = a[2] + dz/2.0;
    return suthbass_alpha(c, z, freq)*ds;
}
/* GeoAc_Jacobian: 3DStratified.cpp:410-428 */
static double s3_jacobian(orc_ctx* c, int k){
    const src_3d* S = &c->S3; const double* y = ROW(c, k);
    double z = y[2], z0 = S->src_loc[2];
    double nu[3] = { S->nu0_xy[0], S->nu0_xy[1], y[3] };
    double cc = atm_c(c, z), uu = atm_u(c, z), vv = atm_v(c, z);
    double nu_mag = (atm_c(c, z0) - nu[0]*uu - nu[1]*vv)/cc;
    double c_prop[3] = { cc*nu[0]/nu_mag + uu, cc*nu[1]/nu_mag + vv, cc*nu[2]/nu_mag + 0.0 };
    double c_prop_mag = sqrt(pow(c_prop[0],2) + pow(c_prop[1],2) + pow(c_prop[2],2));
    double dxds = c_prop[0]/c_prop_mag, dyds = c_prop[1]/c_prop_mag, dzds = c_prop[2]/c_prop_mag;
    double dxdtheta = y[4], dydtheta = y[5], dzdtheta = y[6];
    double dxdphi = y[8], dydphi = y[9], dzdphi = y[10];
    return dxds*(dydtheta*dzdphi - dydphi*dzdtheta)
         - dxdtheta*(dyds*dzdphi - dzds*dydphi)
         + dxdphi*(dyds*dzdtheta - dzds*dydtheta);
}
/* GeoAc_Amplitude: 3DStratified.cpp:431-451 (nu_mag0 sign slip, c_prop[2] without w: Q4) */
static double s3_amplitude(orc_ctx* c, int k){
    const src_3d* S = &c->S3; const double* y = ROW(c, k);
    double z = y[2], z0 = S->src_loc[2];
    double nu[3] = { S->nu0_xy[0], S->nu0_xy[1], y[3] };
    double cc = atm_c(c, z), uu = atm_u(c, z), vv = atm_v(c, z);
    double c0s = atm_c(c, z0), u0s = atm_u(c, z0), v0s = atm_v(c, z0);
    double nu_mag  = (c0s - nu[0]*uu - nu[1]*vv)/cc;
    double nu_mag0 = 1.0 - (nu[0]*u0s - nu[1]*v0s)/c0s;
    double c_prop[3]  = { cc*nu[0]/nu_mag + uu, cc*nu[1]/nu_mag + vv, cc*nu[2]/nu_mag };
    double c_prop0[3] = { c0s*nu[0]/nu_mag0 + u0s, c0s*nu[1]/nu_mag0 + v0s, c0s*sqrt(1.0 - pow(nu[0]/nu_mag0,2) - pow(nu[1]/nu_mag0,2)) };
    double c_prop_mag  = sqrt(pow(c_prop[0],2) + pow(c_prop[1],2) + pow(c_prop[2],2));
    double c_prop_mag0 = sqrt(pow(c_prop0[0],2) + pow(c_prop0[1],2) + pow(c_prop0[2],2));
    double D = s3_jacobian(c, k);
    double Amp_Num = atm_rho(c, z) * nu_mag * pow(atm_c(c, z),3) * c_prop_mag0 * cos(c->theta);
    double Amp_Den = atm_rho(c, z0) * nu_mag0 * pow(atm_c(c, z0),3) * c_prop_mag * D;
    return 1.0/(4.0*Pi)*sqrt(fabs(Amp_Num/Amp_Den));
}
/* ApproximateIntercept + SetReflectionConditions: 3DStratified.cpp:136-186 (quadratic term kept) */
static void s3_reflect(orc_ctx* c, int k){
    const src_3d* S = &c->S3;
    double prev[18];
    const double* yk = ROW(c, k); const double* ykm = ROW(c, k-1); const double* ykmm = ROW(c, k-2);
    double zg = c->z_grnd;
    double dz_k = yk[2] - ykm[2];
    double dz_grnd = ykm[2] - zg;
    for(int i = 0; i < c->EqCnt; i++)
        prev[i] = ykm[i] + (ykm[i] - yk[i])/dz_k*dz_grnd
                + 1.0/2.0*(yk[i] + ykmm[i] - 2.0*ykm[i])/pow(dz_k,2.0)*pow(dz_grnd,2.0);
    double cg = atm_c(c, zg);
    double dnuz_ds = - 1.0/cg * (S->c0/cg * atm_c_diff(c, zg)
                                 + S->nu0_xy[0] * atm_u_diff(c, zg)
                                 + S->nu0_xy[1] * atm_v_diff(c, zg)
                                 + prev[3] * 0.0);
    double* y0 = ROW(c, 0);
    for(int i = 0; i < c->EqCnt; i++){
        if(i == 3 || i == 6 || i == 10) y0[i] = -prev[i];
        else if(i == 7 || i == 11) y0[i] = -prev[i] + 2.0*dnuz_ds*prev[i-1]/(cg/S->c0*prev[3]);
        else y0[i] = prev[i];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* 2-D effective-sound-speed set: GeoAc.EquationSets.2DStratified.cpp                           */
/* ------------------------------------------------------------------------------------------ */
static double s2_ceff(orc_ctx* c, double z){ return atm_c(c, z) + atm_u(c, z)*cos(c->phi) + atm_v(c, z)*sin(c->phi); }
/* SetInitialConditions: 2DStratified.cpp:38-68 */
static void s2_set_ic(orc_ctx* c, double r0, double z0){
    double* y = ROW(c, 0);
    c->S2.c_eff_0 = s2_ceff(c, z0);
    for(int i = 0; i < c->EqCnt; i++){
        switch(i){
            case 0: y[i] = r0; break;
            case 1: y[i] = z0; break;
            case 2: y[i] = sin(c->theta); break;
            case 5: y[i] = cos(c->theta); break;
            default: y[i] = 0.0;
        }
    }
}
/* UpdateSources: 2DStratified.cpp:135-147 */
static void s2_update_sources(orc_ctx* c, const double* y){
    double z = y[1];
    c->S2.c_eff = s2_ceff(c, z);
    c->S2.c_eff_diff = atm_c_diff(c, z) + atm_u_diff(c, z)*cos(c->phi) + atm_v_diff(c, z)*sin(c->phi);
    if(c->CalcAmp)
        c->S2.c_eff_ddiff = atm_c_ddiff(c, z) + atm_u_ddiff(c, z)*cos(c->phi) + atm_v_ddiff(c, z)*sin(c->phi);
}
/* EvalSrcEq: 2DStratified.cpp:152-181 */
static double s2_eval_src_eq(const orc_ctx* c, const double* y, int q){
    double nu_z = y[2], dzt = y[4], mu_z = y[5];      /* reads past E=3 rows like the reference; rows have stride 18 here */
    double cc = c->S2.c_eff, c0 = c->S2.c_eff_0, dc = c->S2.c_eff_diff, ddc = c->S2.c_eff_ddiff;
    switch(q){
        case 0: return cc/c0*cos(c->theta);
        case 1: return cc/c0*nu_z;
        case 2: return -c0/pow(cc,2)*dc;
        case 3: return dc*dzt/c0*cos(c->theta) - cc/c0*sin(c->theta);
        case 4: return dc*dzt/c0*nu_z + cc/c0*mu_z;
        default: return (2*pow(dc/cc,2) - ddc/cc)*c0/cc*dzt;
    }
}
/* BreakCheck / GroundCheck: 2DStratified.cpp:194-212 */
static int s2_break_check(const orc_ctx* c, int k){
    const double* y = ROW(c, k);
    int check = 0;
    if(y[1] > c->vert_limit) check = 1;
    if(y[0] > c->range_limit) check = 1;
    return check;
}
static int s2_ground_check(const orc_ctx* c, int k){ return ROW(c, k)[1] < c->z_grnd; }
/* TravelTime segment: 2DStratified.cpp:217-247 */
static double s2_tt_seg(orc_ctx* c, int n){
    const double* a = ROW(c, n); const double* b = ROW(c, n+1);
    double dr = b[0] - a[0], dz = b[1] - a[1];
    double z_avg = a[1] + dz/2.0;
    double c_eff = s2_ceff(c, z_avg);
    double ds = sqrt(pow(dr,2) + pow(dz,2));
    return ds/c_eff;
}
/* SB_Atten segment: 2DStratified.cpp:252-286 */
static double s2_att_seg(orc_ctx* c, int n, double freq){
    const double* a = ROW(c, n); const double* b = ROW(c, n+1);
    double dr = b[0] - a[0], dz = b[1] - a[1];
    double ds = sqrt(dr*dr + dz*dz);
    double z = a[1] + dz/2.0;
    return suthbass_alpha(c, z, freq)*ds;
}
/* Jacobian / Amplitude: 2DStratified.cpp:291-313 */
static double s2_jacobian(orc_ctx* c, int k){
    const double* y = ROW(c, k);
    double r = y[0], z = y[1];
    double drds = atm_c(c, z)/c->S2.c_eff_0*cos(c->theta);
    double dzds = atm_c(c, z)/c->S2.c_eff_0*y[2];
    return r*(drds*y[4] - dzds*y[3]);
}
static double s2_amplitude(orc_ctx* c, int k){
    double z = ROW(c, k)[1];
    double D = s2_jacobian(c, k);
    double Amp_Num = atm_rho(c, z)*atm_c(c, z)*cos(c->theta);
    double Amp_Den = atm_rho(c, c->z_grnd)*c->S2.c_eff_0*D;
    return 1.0/(4.0*Pi)*sqrt(fabs(Amp_Num/Amp_Den));
}
/* ApproximateIntercept + SetReflectionConditions: 2DStratified.cpp:74-117 */
static void s2_reflect(orc_ctx* c, int k){
    double prev[18];
    const double* yk = ROW(c, k); const double* ykm = ROW(c, k-1); const double* ykmm = ROW(c, k-2);
    double zg = c->z_grnd;
    double dz_k = yk[1] - ykm[1];
    double dz_grnd = ykm[1] - zg;
    for(int i = 0; i < c->EqCnt; i++)
        prev[i] = ykm[i] + (ykm[i] - yk[i])/dz_k*dz_grnd
                + 1.0/2.0*(yk[i] + ykmm[i] - 2.0*ykm[i])/pow(dz_k,2.0)*pow(dz_grnd,2.0);
    double c_eff_diff = atm_c_diff(c, zg) + atm_u_diff(c, zg)*cos(c->phi) + atm_v_diff(c, zg)*sin(c->phi);
    double dnuz_ds = - c->S2.c_eff_0/pow(atm_c(c, zg),2)*c_eff_diff;
    double* y0 = ROW(c, 0);
    for(int i = 0; i < c->EqCnt; i++){
        if(i == 0 || i == 3) y0[i] = prev[i];
        else if(i == 1) y0[i] = zg;
        else if(i == 2 || i == 4) y0[i] = -prev[i];
        else y0[i] = -prev[i] + 2.0*dnuz_ds*prev[4]/(atm_c(c, zg)/c->S2.c_eff_0*prev[2]);
    }
}

#include "geoac_oracle_rngdep.inc.c"
#include "geoac_oracle_globalrd.inc.c"

/* ------------------------------------------------------------------------------------------ */
/* dispatch by equation set                                                                     */
/* ------------------------------------------------------------------------------------------ */
static void update_sources(orc_ctx* c, const double* y){
    if(c->eqset == GEOAC_EQ_GLOBAL) g_update_sources(c, y);
    else if(c->eqset == GEOAC_EQ_3D) s3_update_sources(c, y);
    else if(c->eqset == GEOAC_EQ_3D_RNGDEP) rd_update_sources(c, y);
    else if(c->eqset == GEOAC_EQ_GLOBAL_RNGDEP) grd_update_sources(c, y);
    else s2_update_sources(c, y);
}
static double eval_src_eq(const orc_ctx* c, const double* y, int q){
    if(c->eqset == GEOAC_EQ_GLOBAL || c->eqset == GEOAC_EQ_GLOBAL_RNGDEP) return g_eval_src_eq(c, y, q);   /* GlobalRngDep.cpp:390-458: same text */
    if(c->eqset == GEOAC_EQ_3D) return s3_eval_src_eq(c, y, q);
    if(c->eqset == GEOAC_EQ_3D_RNGDEP) return rd_eval_src_eq(c, y, q);
    return s2_eval_src_eq(c, y, q);
}
static double step_ds(const orc_ctx* c, const double* y){
    if(c->eqset == GEOAC_EQ_GLOBAL || c->eqset == GEOAC_EQ_GLOBAL_RNGDEP) return set_ds(c, y[0] - (c->r_earth + c->z_grnd));
    if(c->eqset == GEOAC_EQ_3D || c->eqset == GEOAC_EQ_3D_RNGDEP) return set_ds(c, y[2] - c->z_grnd);
    return set_ds(c, y[1] - c->z_grnd);
}
static int break_check(const orc_ctx* c, int k){
    if(c->eqset == GEOAC_EQ_GLOBAL) return g_break_check(c, k);
    if(c->eqset == GEOAC_EQ_3D) return s3_break_check(c, k);
    if(c->eqset == GEOAC_EQ_3D_RNGDEP) return rd_break_check(c, k);
    if(c->eqset == GEOAC_EQ_GLOBAL_RNGDEP) return grd_break_check(c, k);
    return s2_break_check(c, k);
}
static int ground_check(const orc_ctx* c, int k){
    if(c->eqset == GEOAC_EQ_GLOBAL || c->eqset == GEOAC_EQ_GLOBAL_RNGDEP) return g_ground_check(c, k);
    if(c->eqset == GEOAC_EQ_3D) return s3_ground_check(c, k);
    if(c->eqset == GEOAC_EQ_3D_RNGDEP) return rd_ground_check(c, k);
    return s2_ground_check(c, k);
}
static double tt_seg(orc_ctx* c, int n){
    if(c->eqset == GEOAC_EQ_GLOBAL) return g_tt_seg(c, n);
    if(c->eqset == GEOAC_EQ_3D) return s3_tt_seg(c, n);
    if(c->eqset == GEOAC_EQ_3D_RNGDEP) return rd_tt_seg(c, n);
    if(c->eqset == GEOAC_EQ_GLOBAL_RNGDEP) return grd_tt_seg(c, n);
    return s2_tt_seg(c, n);
}
static double att_seg(orc_ctx* c, int n, double f){
    if(c->eqset == GEOAC_EQ_GLOBAL) return g_att_seg(c, n, f);
    if(c->eqset == GEOAC_EQ_3D) return s3_att_seg(c, n, f);
    if(c->eqset == GEOAC_EQ_3D_RNGDEP) return rd_att_seg(c, n, f);
    if(c->eqset == GEOAC_EQ_GLOBAL_RNGDEP) return grd_att_seg(c, n, f);
    return s2_att_seg(c, n, f);
}
static double jacobian(orc_ctx* c, int k){
    if(c->eqset == GEOAC_EQ_GLOBAL) return g_jacobian(c, k);
    if(c->eqset == GEOAC_EQ_3D) return s3_jacobian(c, k);
    if(c->eqset == GEOAC_EQ_3D_RNGDEP) return rd_jacobian(c, k);
    if(c->eqset == GEOAC_EQ_GLOBAL_RNGDEP) return grd_jacobian(c, k);
    return s2_jacobian(c, k);
}
static double amplitude(orc_ctx* c, int k){
    if(c->eqset == GEOAC_EQ_GLOBAL) return g_amplitude(c, k);
    if(c->eqset == GEOAC_EQ_3D) return s3_amplitude(c, k);
    if(c->eqset == GEOAC_EQ_3D_RNGDEP) return rd_amplitude(c, k);
    if(c->eqset == GEOAC_EQ_GLOBAL_RNGDEP) return grd_amplitude(c, k);
    return s2_amplitude(c, k);
}
static void reflect(orc_ctx* c, int k){
    if(c->eqset == GEOAC_EQ_GLOBAL) g_reflect(c, k);
    else if(c->eqset == GEOAC_EQ_3D) s3_reflect(c, k);
    else if(c->eqset == GEOAC_EQ_3D_RNGDEP) rd_reflect(c, k);
    else if(c->eqset == GEOAC_EQ_GLOBAL_RNGDEP) grd_reflect(c, k);
    else s2_reflect(c, k);
}

/* GeoAc_SetEqCnt: GeoAc.Interface.cpp:21-41 with SetSystem of each set
 * (2D: dim 2 strat -> 3/6; 3D: dim 3 strat -> 4/12; Global: dim 3 non-strat -> 6/18) */
static void configure(orc_ctx* c, int calc_amp){
    c->CalcAmp = calc_amp ? 1 : 0;
    if(c->eqset == GEOAC_EQ_2D) c->EqCnt = calc_amp ? 6 : 3;
    else if(c->eqset == GEOAC_EQ_3D) c->EqCnt = calc_amp ? 12 : 4;
    else c->EqCnt = calc_amp ? 18 : 6;                          /* Global and the RngDep sets: dim 3, not stratified */
}

static int64_t step_limit(const orc_ctx* c){ return (int64_t)(c->ray_limit * (int)(1.0/(c->ds_min*10))); }   /* Solver.cpp:14 */

static void ensure_solution(orc_ctx* c){
    int64_t rows = step_limit(c);
    if(c->sol && c->sol_rows == rows) return;
    free(c->sol);
    c->sol = (double*)calloc((size_t)rows * SOLSTRIDE, sizeof(double));   /* lazily committed by the OS */
    c->sol_rows = rows;
}

/* GeoAc_Propagate_RK4: GeoAc.Solver.cpp:12-72 */
static int propagate_rk4(orc_ctx* c, int* check){
    int k = 0;
    int64_t limit = step_limit(c);
    int E = c->EqCnt;
    double ds;
    double temp0[SOLSTRIDE] = {0}, temp1[SOLSTRIDE], temp2[SOLSTRIDE], temp3[SOLSTRIDE], temp4[SOLSTRIDE];
    double partial1[SOLSTRIDE] = {0}, partial2[SOLSTRIDE] = {0}, partial3[SOLSTRIDE] = {0};
    *check = 0;
    for(k = 0; k < (limit - 1); k++){
        double* yk = ROW(c, k); double* yn = ROW(c, k+1);
        for(int i = 0; i < E; i++) temp0[i] = yk[i];
        update_sources(c, temp0);
        ds = step_ds(c, temp0);
        for(int i = 0; i < E; i++){ temp1[i] = ds*eval_src_eq(c, temp0, i); partial1[i] = yk[i] + temp1[i]/2.0; }
        update_sources(c, partial1);
        for(int i = 0; i < E; i++){ temp2[i] = ds*eval_src_eq(c, partial1, i); partial2[i] = yk[i] + temp2[i]/2.0; }
        update_sources(c, partial2);
        for(int i = 0; i < E; i++){ temp3[i] = ds*eval_src_eq(c, partial2, i); partial3[i] = yk[i] + temp3[i]; }
        update_sources(c, partial3);
        for(int i = 0; i < E; i++){
            temp4[i] = ds*eval_src_eq(c, partial3, i);
            yn[i] = yk[i] + temp1[i]/6.0 + temp2[i]/3.0 + temp3[i]/3.0 + temp4[i]/6.0;
        }
        if(break_check(c, k+1)){ *check = 1; break; }
        if(ground_check(c, k+1)){ *check = 0; break; }
    }
    return k + 1;
}

static void apply_cfg(orc_ctx* c, const ref_fan_cfg* cfg){
    c->z_grnd = cfg->z_grnd;
    c->tweak_abs = cfg->tweak_abs;
    if(cfg->vert_limit == cfg->vert_limit)   c->vert_limit  = cfg->vert_limit;
    if(cfg->range_limit == cfg->range_limit) c->range_limit = cfg->range_limit;
    if(cfg->xy_limits[0] == cfg->xy_limits[0]) c->x_min_limit = cfg->xy_limits[0];
    if(cfg->xy_limits[1] == cfg->xy_limits[1]) c->x_max_limit = cfg->xy_limits[1];
    if(cfg->xy_limits[2] == cfg->xy_limits[2]) c->y_min_limit = cfg->xy_limits[2];
    if(cfg->xy_limits[3] == cfg->xy_limits[3]) c->y_max_limit = cfg->xy_limits[3];
    int calc = cfg->calc_amp != 0;
    if(cfg->mode & GEOAC_MODE_WRITE_CAUSTICS) calc = 1;
    configure(c, calc);
    ensure_solution(c);
}

static void set_launch(orc_ctx* c, double theta_deg, double phi_deg){
    c->theta = theta_deg*Pi/180.0;                  /* GeoAcGlobal_main.cpp:244 */
    c->phi = Pi/2.0 - phi_deg*Pi/180.0;             /* :245 */
}

static void set_ic(orc_ctx* c, const ref_fan_cfg* cfg){
    if(c->eqset == GEOAC_EQ_GLOBAL_RNGDEP){
        double z_src = DMAX(cfg->src[0], c->z_grnd);              /* GeoAcGlobal.RngDep_main.cpp:177 */
        grd_set_ic(c, z_src, cfg->src[1]*Pi/180.0, cfg->src[2]*Pi/180.0);
    } else if(c->eqset == GEOAC_EQ_GLOBAL){
        double z_src = DMAX(cfg->src[0], c->z_grnd);
        g_set_ic(c, z_src, cfg->src[1]*Pi/180.0, cfg->src[2]*Pi/180.0);
    } else if(c->eqset == GEOAC_EQ_3D){
        double z_src = DMAX(c->z_grnd, cfg->src[2]);
        s3_set_ic(c, cfg->src[0], cfg->src[1], z_src);
    } else if(c->eqset == GEOAC_EQ_3D_RNGDEP){
        double z_src = DMAX(c->z_grnd, cfg->src[2]);          /* GeoAc3D.RngDep_main.cpp:165 */
        rd_set_ic(c, cfg->src[0], cfg->src[1], z_src);
    } else {
        double z_src = DMAX(cfg->src[0], c->z_grnd);
        s2_set_ic(c, 0.0, z_src);
    }
}

/* the fan / bounce / post-pass loops: GeoAcGlobal_main.cpp:241-325, GeoAc3D_main.cpp:226-307, GeoAc2D_main.cpp:170-232 */
int64_t orc_fan(orc_ctx* c, const ref_fan_cfg* cfg, int n, const double* theta_deg, const double* phi_deg,
                double* rec, double* smp, int64_t smp_cap, int64_t* n_smp){
    apply_cfg(c, cfg);
    const int CalcAmp = c->CalcAmp;
    const int is2d = (c->eqset == GEOAC_EQ_2D), isrd = (c->eqset == GEOAC_EQ_3D_RNGDEP), is3d = (c->eqset == GEOAC_EQ_3D) || isrd;
    const int WriteRays = is2d ? 1 : ((cfg->mode & GEOAC_MODE_WRITE_RAYS) != 0);   /* GeoAc2D always writes raypaths */
    const int WriteCaustics = (cfg->mode & GEOAC_MODE_WRITE_CAUSTICS) != 0;
    const int bounces = cfg->bounces;
    const double freq = cfg->freq;
    const int hidx = is2d ? 1 : (is3d ? 2 : 0);         /* index of the height component */
    const int isgrd = (c->eqset == GEOAC_EQ_GLOBAL_RNGDEP);
    const double hoff = (c->eqset == GEOAC_EQ_GLOBAL || isgrd) ? c->r_earth : 0.0;

    memset(rec, 0, sizeof(double) * (size_t)n * (bounces + 1) * GEOAC_REC_STRIDE);
    int64_t total_steps = 0, ns = 0;
    int k = 0, BreakCheck = 0;

    for(int i = 0; i < n; i++){
        double theta = theta_deg[i], phi = phi_deg[i];
        set_launch(c, theta, phi);
        set_ic(c, cfg);
        double travel_time_sum = 0.0, attenuation = 0.0, h_max = 0.0, D = 0.0, D_prev = 0.0;

        for(int bnc_cnt = 0; bnc_cnt <= bounces; bnc_cnt++){
            double* R = rec + ((size_t)i * (bounces + 1) + bnc_cnt) * GEOAC_REC_STRIDE;
            k = propagate_rk4(c, &BreakCheck);
            total_steps += k;
            R[GEOAC_REC_STEPS] = k;
            R[GEOAC_REC_BROKE] = BreakCheck ? 1.0 : 0.0;

            if(WriteRays || WriteCaustics){
                if(WriteCaustics) D_prev = jacobian(c, 1);
                for(int m = 1; m < k; m++){
                    if(WriteCaustics && !is2d) D = jacobian(c, m);
                    travel_time_sum += tt_seg(c, m-1);
                    attenuation += att_seg(c, m-1, freq);
                    if(WriteCaustics && is2d) D = jacobian(c, m);
                    const double* y = ROW(c, m);
                    if(WriteRays && m % 25 == 0){
                        if(smp && ns < smp_cap){
                            double* S = smp + ns * GEOAC_SMP_STRIDE;
                            S[GEOAC_SMP_RAY] = i; S[GEOAC_SMP_LEG] = bnc_cnt; S[GEOAC_SMP_M] = m; S[GEOAC_SMP_KIND] = 0;
                            double amp_db = CalcAmp ? 20.0*log10(amplitude(c, m)) : 0.0;
                            if(is2d){        S[4] = y[0]; S[5] = DMAX(y[1], 0.0); S[6] = amp_db; S[7] = -attenuation; S[8] = travel_time_sum; S[9] = 0; }
                            else if(is3d){   S[4] = y[0]; S[5] = y[1]; S[6] = DMAX(y[2], 0.0); S[7] = amp_db; S[8] = -attenuation; S[9] = travel_time_sum; }
                            else {           S[4] = y[0] - c->r_earth; S[5] = y[1]*180.0/Pi; S[6] = y[2]*180.0/Pi; S[7] = amp_db; S[8] = -attenuation; S[9] = travel_time_sum; }
                        }
                        ns++;
                    }
                    if(WriteCaustics && D*D_prev < 0.0){
                        if(smp && ns < smp_cap){
                            double* S = smp + ns * GEOAC_SMP_STRIDE;
                            S[GEOAC_SMP_RAY] = i; S[GEOAC_SMP_LEG] = bnc_cnt; S[GEOAC_SMP_M] = m; S[GEOAC_SMP_KIND] = 1;
                            if(is2d){        S[4] = y[0]; S[5] = y[1]; S[6] = travel_time_sum; S[7] = 0; }
                            else if(isrd){   S[4] = y[0]; S[5] = y[1]; S[6] = y[2]; S[7] = 0.0; S[8] = travel_time_sum; S[9] = 0; }   /* GeoAc3D.RngDep_main.cpp:279-284: raw z, a 0.0 column */
                            else if(is3d){   S[4] = y[0]; S[5] = y[1]; S[6] = DMAX(y[2], 0.0); S[7] = travel_time_sum; }
                            else {           S[4] = y[0] - c->r_earth; S[5] = y[1]*180.0/Pi; S[6] = y[2]*180.0/Pi; S[7] = travel_time_sum; }
                            if(!isrd){ S[8] = 0; S[9] = 0; }
                        }
                        ns++;
                    }
                    if(WriteCaustics) D_prev = D;
                }
            } else {
                double tt = 0.0, at = 0.0;                                  /* GeoAc_TravelTime / GeoAc_SB_Atten start from 0 */
                for(int m = 0; m < k; m++) tt += tt_seg(c, m);
                travel_time_sum += tt;
                for(int m = 0; m < k; m++) at += att_seg(c, m, freq);
                attenuation += at;
            }
            R[GEOAC_REC_TTIME] = travel_time_sum;
            R[GEOAC_REC_ATTEN] = attenuation;

            if(BreakCheck) break;
            if(isrd || isgrd) h_max = 0.0;                                           /* turning height per leg in the RngDep mains (Q8) */
            for(int m = 0; m < k; m++) h_max = DMAX(h_max, ROW(c, m)[hidx] - hoff);

            const double* yk = ROW(c, k);
            R[GEOAC_REC_VALID] = 1.0;
            R[GEOAC_REC_TURN]  = h_max;
            if(isgrd){                                                      /* GeoAcGlobal.RngDep_main.cpp:307-313: no leading minus (Q10) */
                double lat_src = cfg->src[1], lon_src = cfg->src[2];
                double z_src = DMAX(cfg->src[0], c->z_grnd);
                double GC_Dist1 = pow(sin((yk[1] - lat_src*Pi/180.0)/2.0),2);
                double GC_Dist2 = cos(lat_src*Pi/180.0) * cos(yk[1]) * pow(sin((yk[2] - lon_src*Pi/180.0)/2.0),2);
                double inclination = asin(gg_c(c->GG, yk[0], yk[1], yk[2]) / gg_c(c->GG, c->r_earth + z_src, lat_src*Pi/180.0, lon_src*Pi/180.0) * yk[3]) * 180.0 / Pi;
                double back_az = 90.0 - atan2(-yk[4], -yk[5]) * 180.0 / Pi;
                if(back_az < -180.0) back_az += 360.0;
                if(back_az >  180.0) back_az -= 360.0;
                R[GEOAC_REC_INCL] = inclination; R[GEOAC_REC_BACKAZ] = back_az;
                R[GEOAC_REC_RANGE] = 2.0 * c->r_earth * asin(sqrt(GC_Dist1+GC_Dist2));
            } else if(c->eqset == GEOAC_EQ_GLOBAL){                         /* GeoAcGlobal_main.cpp:296-302 */
                double lat_src = cfg->src[1], lon_src = cfg->src[2];
                double z_src = DMAX(cfg->src[0], c->z_grnd);
                double GC_Dist1 = pow(sin((yk[1] - lat_src*Pi/180.0)/2.0),2);
                double GC_Dist2 = cos(lat_src*Pi/180.0) * cos(yk[1]) * pow(sin((yk[2] - lon_src*Pi/180.0)/2.0),2);
                double inclination = - asin(atm_c(c, yk[0]) / atm_c(c, c->r_earth + z_src) * yk[3]) * 180.0 / Pi;
                double back_az = 90.0 - atan2(-yk[4], -yk[5]) * 180.0 / Pi;
                if(back_az < -180.0) back_az += 360.0;
                if(back_az >  180.0) back_az -= 360.0;
                R[GEOAC_REC_INCL] = inclination; R[GEOAC_REC_BACKAZ] = back_az;
                R[GEOAC_REC_RANGE] = 2.0 * c->r_earth * asin(sqrt(GC_Dist1+GC_Dist2));
            } else if(isrd){                                                /* GeoAc3D.RngDep_main.cpp:298-301 */
                double z_src = DMAX(c->z_grnd, cfg->src[2]);
                double inclination = - asin(g3_c(c->G3, yk[0], yk[1], c->z_grnd) / g3_c(c->G3, cfg->src[0], cfg->src[1], z_src) * yk[5]) * 180.0 / Pi;
                double back_az = 90.0 - atan2(-yk[4], -yk[3]) * 180.0 / Pi;
                while(back_az < -180.0) back_az += 360.0;
                while(back_az >  180.0) back_az -= 360.0;
                R[GEOAC_REC_INCL] = inclination; R[GEOAC_REC_BACKAZ] = back_az;
                R[GEOAC_REC_RANGE] = sqrt(yk[0]*yk[0] + yk[1]*yk[1]);
            } else if(is3d){                                                /* GeoAc3D_main.cpp:281-284 */
                double z_src = DMAX(c->z_grnd, cfg->src[2]);
                double back_az = phi + 180.0;
                double inclination = - asin(atm_c(c, c->z_grnd) / atm_c(c, z_src) * yk[3]) * 180.0 / Pi;
                while(back_az > 180.0)  back_az -= 360.0;
                while(back_az < -180.0) back_az += 360.0;
                R[GEOAC_REC_INCL] = inclination; R[GEOAC_REC_BACKAZ] = back_az;
                R[GEOAC_REC_RANGE] = sqrt(yk[0]*yk[0] + yk[1]*yk[1]);
            } else {                                                        /* GeoAc2D_main.cpp:216-226 */
                R[GEOAC_REC_INCL] = -theta; R[GEOAC_REC_BACKAZ] = 0.0; R[GEOAC_REC_RANGE] = yk[0];
            }
            if(CalcAmp){
                R[GEOAC_REC_AMP]   = amplitude(c, k);
                R[GEOAC_REC_JACOB] = jacobian(c, k);
            }
            for(int e = 0; e < c->EqCnt; e++) R[GEOAC_REC_STATE + e] = yk[e];

            reflect(c, k);
        }
        /* GeoAc_ClearSolutionArray(solution,k): rows 0..k-1 zeroed (GeoAcGlobal_main.cpp:322) */
        for(int m = 0; m < k; m++) memset(ROW(c, m), 0, sizeof(double) * (size_t)c->EqCnt);
    }
    if(n_smp) *n_smp = ns;
    return total_steps;
}

/* ------------------------------------------------------------------------------------------ */
/* probes                                                                                       */
/* ------------------------------------------------------------------------------------------ */
void orc_atmo_probe(orc_ctx* c, int n, const double* x, double* o, double* rho_out){
    for(int i = 0; i < n; i++){
        double r = x[i];
        o[9*i+0] = atm_c(c, r); o[9*i+1] = atm_c_diff(c, r); o[9*i+2] = atm_c_ddiff(c, r);
        o[9*i+3] = atm_u(c, r); o[9*i+4] = atm_u_diff(c, r); o[9*i+5] = atm_u_ddiff(c, r);
        o[9*i+6] = atm_v(c, r); o[9*i+7] = atm_v_diff(c, r); o[9*i+8] = atm_v_ddiff(c, r);
        rho_out[i] = atm_rho(c, r);
    }
}
void orc_absorption_probe(orc_ctx* c, int n, const double* x, const double* f, double zg, double tweak, double* out){
    double zg0 = c->z_grnd, tw0 = c->tweak_abs;
    c->z_grnd = zg; c->tweak_abs = tweak;
    for(int i = 0; i < n; i++) out[i] = suthbass_alpha(c, x[i], f[i]);
    c->z_grnd = zg0; c->tweak_abs = tw0;
}
int orc_tables(orc_ctx* c, int cap, double* x, double* T, double* u, double* v, double* rho,
               double* sT, double* su, double* sv, double* srho){
    int n = c->n;
    if(n > cap) return -n;
    size_t b = sizeof(double) * (size_t)n;
    memcpy(x, c->x, b); memcpy(T, c->T, b); memcpy(u, c->u, b); memcpy(v, c->v, b); memcpy(rho, c->rho, b);
    memcpy(sT, c->sT, b); memcpy(su, c->su, b); memcpy(sv, c->sv, b); memcpy(srho, c->srho, b);
    return n;
}
int orc_trace_leg0(orc_ctx* c, const ref_fan_cfg* cfg, double theta_deg, double phi_deg, int max_rows, double* out, int* E){
    apply_cfg(c, cfg);
    set_launch(c, theta_deg, phi_deg);
    set_ic(c, cfg);
    int chk;
    int k = propagate_rk4(c, &chk);
    *E = c->EqCnt;
    for(int m = 0; m <= k && m < max_rows; m++)
        for(int e = 0; e < c->EqCnt; e++) out[(size_t)m*c->EqCnt + e] = ROW(c, m)[e];
    for(int m = 0; m < k; m++) memset(ROW(c, m), 0, sizeof(double) * (size_t)c->EqCnt);
    return chk ? -k : k;
}

/* ------------------------------------------------------------------------------------------ */
/* range-dependent API                                                                          */
/* ------------------------------------------------------------------------------------------ */
int orc_load_grid(orc_ctx* c, const char* prefix, const char* locx, const char* locy, const char* format, double z_grnd_at_load){
    if(c->eqset == GEOAC_EQ_GLOBAL_RNGDEP){
        gg_free(c->GG);
        c->GG = gg_load(prefix, locx, locy, format, z_grnd_at_load, c->r_earth);
        if(!c->GG) return -1;
        /* GeoAc_SetPropRegion: G2S_GlobalMultiDimSpline3D.cpp:25-33 */
        c->vert_limit = c->GG->r_max;
        c->x_min_limit = c->GG->t_min; c->x_max_limit = c->GG->t_max;
        c->y_min_limit = c->GG->p_min; c->y_max_limit = c->GG->p_max;
        return c->GG->nr;
    }
    if(c->eqset != GEOAC_EQ_3D_RNGDEP) return -10;
    g3_free(c->G3);
    c->G3 = g3_load(prefix, locx, locy, format, z_grnd_at_load);
    if(!c->G3) return -1;
    /* GeoAc_SetPropRegion: G2S_MultiDimSpline3D.cpp:25-33 */
    c->vert_limit = c->G3->z_max;
    c->x_min_limit = c->G3->x_min; c->x_max_limit = c->G3->x_max;
    c->y_min_limit = c->G3->y_min; c->y_max_limit = c->G3->y_max;
    return c->G3->nz;
}
void orc_grid_dims(orc_ctx* c, int* nx, int* ny, int* nz){
    if(c->GG){ *nx = c->GG->nt; *ny = c->GG->np; *nz = c->GG->nr; return; }
    *nx = c->G3->nx; *ny = c->G3->ny; *nz = c->G3->nz;
}
void orc_grid_centre(orc_ctx* c, double* lat_deg, double* lon_deg){      /* GeoAcGlobal.RngDep_main.cpp:135-137 */
    *lat_deg = (c->GG->tv[0] + c->GG->tv[c->GG->nt-1])/2.0 * 180.0/Pi;
    *lon_deg = (c->GG->pv[0] + c->GG->pv[c->GG->np-1])/2.0 * 180.0/Pi;
}

/* out30[i]: AllOrder2 of T, u, v (10 each); api8[i]: c, rho, u, v, c_diff(z), u_diff(z), v_diff(z), c_diff(x) scalar API values */
void orc_grid_probe(orc_ctx* c, int n, const double* x, const double* y, const double* z, double* out30, double* api8){
    if(c->GG){                                                          /* (x, y, z) = (r, lat, lon); api8 = c, rho, u, v, dc/dr, du/dr, dv/dr, dc/dlat */
        struct gridg* Q = c->GG;
        for(int i = 0; i < n; i++){
            gg_eval_all(Q, x[i], y[i], z[i], &Q->Temp, 1, out30 + 30*i);
            gg_eval_all(Q, x[i], y[i], z[i], &Q->Windu, 1, out30 + 30*i + 10);
            gg_eval_all(Q, x[i], y[i], z[i], &Q->Windv, 1, out30 + 30*i + 20);
            double* a = api8 + 8*i;
            a[0] = gg_c(Q, x[i], y[i], z[i]); a[1] = gg_rho(Q, x[i], y[i], z[i]); a[2] = gg_u(Q, x[i], y[i], z[i]); a[3] = gg_v(Q, x[i], y[i], z[i]);
            a[4] = gg_c_diff(Q, x[i], y[i], z[i], 0); a[5] = gg_u_diff(Q, x[i], y[i], z[i], 0); a[6] = gg_v_diff(Q, x[i], y[i], z[i], 0);
            a[7] = gg_c_diff(Q, x[i], y[i], z[i], 1);
        }
        return;
    }
    struct grid3d* G = c->G3;
    for(int i = 0; i < n; i++){
        g3_eval_all(G, x[i], y[i], z[i], &G->Temp, 1, out30 + 30*i);
        g3_eval_all(G, x[i], y[i], z[i], &G->Windu, 1, out30 + 30*i + 10);
        g3_eval_all(G, x[i], y[i], z[i], &G->Windv, 1, out30 + 30*i + 20);
        double* a = api8 + 8*i;
        a[0] = g3_c(G, x[i], y[i], z[i]); a[1] = g3_rho(G, x[i], y[i], z[i]); a[2] = g3_u(G, x[i], y[i], z[i]); a[3] = g3_v(G, x[i], y[i], z[i]);
        a[4] = g3_c_diff(G, x[i], y[i], z[i], 2); a[5] = g3_u_diff(G, x[i], y[i], z[i], 2); a[6] = g3_v_diff(G, x[i], y[i], z[i], 2);
        a[7] = g3_c_diff(G, x[i], y[i], z[i], 0);
    }
}
