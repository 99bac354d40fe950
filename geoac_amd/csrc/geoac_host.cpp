// geoac_host.cpp - once-per-job host set-up (see include/geoac_host.h).  Plain C++; part of libgeoac_hip.so.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/geoac_hip.h"
#include "../../include/geoac_host.h"

namespace {
const double kREarth = 6370.0;     // G2S_GlobalSpline1D.cpp:35

bool is_global(int eqset){ return eqset == GEOAC_EQ_GLOBAL || eqset == GEOAC_EQ_GLOBAL_RNGDEP; }

// wind taper towards the ground + m/s -> km/s (G2S_Spline1D.cpp:122-124); z_grnd is 0 when the -prop mains load the profile
double taper(double z, double z_grnd){ return (2.0 / (1.0 + exp(-(z - z_grnd) / 0.2)) - 1.0) / 1000.0; }
}

extern "C" {

int geoac_met_rows(const char* file){
    FILE* fp = fopen(file, "r");
    if(!fp) return -1;
    int rows = 0, ch;
    while((ch = fgetc(fp)) != EOF) if(ch == '\n') rows++;
    fclose(fp);
    return rows;
}

int geoac_met_from_columns(int eqset, int n, const double* z, const double* T, const double* u_ms,
                           const double* v_ms, const double* rho_in,
                           double* x, double* T_out, double* u, double* v, double* rho){
    return geoac_met_from_columns_zg(eqset, 0.0, n, z, T, u_ms, v_ms, rho_in, x, T_out, u, v, rho);
}

int geoac_met_from_columns_zg(int eqset, double z_grnd, int n, const double* z, const double* T, const double* u_ms,
                              const double* v_ms, const double* rho_in,
                              double* x, double* T_out, double* u, double* v, double* rho){
    for(int i = 0; i < n; i++){
        double xi = z[i];
        double w;
        if(is_global(eqset)){
            xi += kREarth;                           // G2S_GlobalSpline1D.cpp:128
            w = taper(xi - kREarth, z_grnd);         // :129 evaluates (r - r_earth - z_grnd)
        } else {
            w = taper(xi, z_grnd);
        }
        x[i] = xi; T_out[i] = T[i]; rho[i] = rho_in[i];
        u[i] = u_ms[i] * w;
        v[i] = v_ms[i] * w;
    }
    return n;
}

int geoac_met_load(const char* file, const char* format, int eqset, int cap,
                   double* x, double* T, double* u, double* v, double* rho){
    return geoac_met_load_zg(file, format, eqset, 0.0, cap, x, T, u, v, rho);
}

int geoac_met_load_zg(const char* file, const char* format, int eqset, double z_grnd, int cap,
                      double* x, double* T, double* u, double* v, double* rho){
    int fmt;
    if(strncmp(format, "zTuvdp", 6) == 0) fmt = 0;
    else if(strncmp(format, "zuvwTdp", 7) == 0) fmt = 1;
    else return -2;
    int rows = geoac_met_rows(file);
    if(rows < 3) return rows < 0 ? rows : -3;
    if(rows > cap) return -4;
    FILE* fp = fopen(file, "r");
    if(!fp) return -1;
    std::vector<double> z(rows), Tc(rows), uc(rows), vc(rows), rc(rows);
    const int ncol = fmt ? 7 : 6;
    for(int i = 0; i < rows; i++){
        double t[7] = {0, 0, 0, 0, 0, 0, 0};
        for(int j = 0; j < ncol; j++) if(fscanf(fp, "%lf", &t[j]) != 1) t[j] = 0.0;
        if(fmt == 0){ z[i] = t[0]; Tc[i] = t[1]; uc[i] = t[2]; vc[i] = t[3]; rc[i] = t[4]; }
        else        { z[i] = t[0]; uc[i] = t[1]; vc[i] = t[2]; Tc[i] = t[4]; rc[i] = t[5]; }
    }
    fclose(fp);
    return geoac_met_from_columns_zg(eqset, z_grnd, rows, z.data(), Tc.data(), uc.data(), vc.data(), rc.data(), x, T, u, v, rho);
}

int geoac_grid_dims(const char* prefix, const char* locx, const char* locy, int* nx, int* ny, int* nz){
    std::string first = std::string(prefix) + "0.met";
    *nx = geoac_met_rows(locx); *ny = geoac_met_rows(locy); *nz = geoac_met_rows(first.c_str());
    return (*nx >= 2 && *ny >= 2 && *nz >= 3) ? 0 : -1;
}

int geoac_grid_load_eq(int eqset, const char* prefix, const char* locx, const char* locy, const char* format, double z_grnd,
                       int nx, int ny, int nz, double* x, double* y, double* z,
                       double* T, double* u, double* v, double* rho){
    const bool glob = (eqset == GEOAC_EQ_GLOBAL_RNGDEP);
    if(!glob && eqset != GEOAC_EQ_3D_RNGDEP) return -3;
    int fmt;
    if(strncmp(format, "zTuvdp", 6) == 0) fmt = 0;
    else if(strncmp(format, "zuvwTdp", 7) == 0) fmt = 1;
    else return -2;
    const double Pi = 3.141592653589793238462643;
    FILE* fp = fopen(locx, "r"); if(!fp) return -1;
    for(int i = 0; i < nx; i++){ if(fscanf(fp, "%lf", &x[i]) != 1) x[i] = 0.0; if(glob) x[i] *= Pi / 180.0; }   // G2S_GlobalMultiDimSpline3D.cpp:143-146
    fclose(fp);
    fp = fopen(locy, "r"); if(!fp) return -1;
    for(int j = 0; j < ny; j++){ if(fscanf(fp, "%lf", &y[j]) != 1) y[j] = 0.0; if(glob) y[j] *= Pi / 180.0; }
    fclose(fp);
    const int ncol = fmt ? 7 : 6;
    for(int i = 0; i < nx; i++) for(int j = 0; j < ny; j++){
        std::string name = std::string(prefix) + std::to_string(i * ny + j) + ".met";       // both sets: first-axis index * second-axis count + second-axis index
        fp = fopen(name.c_str(), "r");
        if(!fp) return -1;
        for(int k = 0; k < nz; k++){
            double t[7] = {0, 0, 0, 0, 0, 0, 0};
            for(int q = 0; q < ncol; q++) if(fscanf(fp, "%lf", &t[q]) != 1) t[q] = 0.0;
            double zz, TT, uu, vv, rr;
            if(fmt == 0){ zz = t[0]; TT = t[1]; uu = t[2]; vv = t[3]; rr = t[4]; }
            else        { zz = t[0]; uu = t[1]; vv = t[2]; TT = t[4]; rr = t[5]; }
            double w;
            if(glob){
                zz += kREarth;                                                           // :172-174: radius, then (r - r_earth - z_grnd) / 0.2
                w = (2.0 / (1.0 + exp(-(zz - kREarth - z_grnd) / 0.2)) - 1.0) / 1000.0;
            } else {
                w = (2.0 / (1.0 + exp(-(zz - z_grnd) / 0.05)) - 1.0) / 1000.0;            // G2S_MultiDimSpline3D.cpp:167-168
            }
            z[k] = zz;
            size_t o = ((size_t)i * ny + j) * nz + k;
            T[o] = TT; u[o] = uu * w; v[o] = vv * w; rho[o] = rr;
        }
        fclose(fp);
    }
    return 0;
}

int geoac_grid_load(const char* prefix, const char* locx, const char* locy, const char* format, double z_grnd,
                    int nx, int ny, int nz, double* x, double* y, double* z,
                    double* T, double* u, double* v, double* rho){
    return geoac_grid_load_eq(GEOAC_EQ_3D_RNGDEP, prefix, locx, locy, format, z_grnd, nx, ny, nz, x, y, z, T, u, v, rho);
}

// cubic of Eval_Spline_f (G2S_Spline1D.cpp:245-281) expanded in powers of t = x - x_k:
//   f = f_k + s_k t + (B - 2A)/h^2 t^2 + (A - B)/h^3 t^3,  A = s_k h - df,  B = -s_{k+1} h + df
// `deriv_form`: store (c0, c1, 2 c2, 6 c3) - the layout the kernels evaluate (value + two derivatives in 6 FMAs)
void geoac_spline_segment_cubic(double x0, double x1, double f0, double f1, double s0, double s1, double* c, int deriv_form){
    long double h = (long double)x1 - (long double)x0;
    long double df = (long double)f1 - (long double)f0;
    long double A = (long double)s0 * h - df, B = -(long double)s1 * h + df;
    long double c2 = (B - 2.0L * A) / (h * h), c3 = (A - B) / (h * h * h);
    c[0] = f0;
    c[1] = s0;
    c[2] = (double)(deriv_form ? 2.0L * c2 : c2);
    c[3] = (double)(deriv_form ? 6.0L * c3 : c3);
}

size_t geoac_grid_table_size(int nx, int ny, int nz){
    const size_t nseg = (size_t)(nz - 1), nn = (size_t)nx * ny;
    return 3 * nseg * nn * 40 + nseg * nn * 16;
}

// the df/dt and df/dp slope systems of the spherical interpolant: interior right-hand sides use (d[i] - d[i+1]) where
// (d[i] - d[i-1]) is meant (G2S_GlobalMultiDimSpline3D.cpp:381, :414; quirk Q12a) - same Thomas sweep otherwise
static void slopes_q12a(int n, const double* x, const double* f, double* slopes){
    std::vector<double> cp(n), dp(n);
    double h0 = x[1] - x[0];
    double b = 2.0 / h0, c = 1.0 / h0, d = 3.0 * (f[1] - f[0]) / (h0 * h0);
    cp[0] = c / b; dp[0] = d / b;
    for(int i = 1; i < n - 1; i++){
        double hl = x[i] - x[i - 1], hr = x[i + 1] - x[i];
        double a = 1.0 / hl;
        b = 2.0 * (1.0 / hl + 1.0 / hr);
        c = 1.0 / hr;
        d = 3.0 * ((f[i] - f[i + 1]) / (hl * hl) + (f[i + 1] - f[i]) / (hr * hr));
        double den = b - cp[i - 1] * a;
        cp[i] = c / den;
        dp[i] = (d - dp[i - 1] * a) / den;
    }
    double hn = x[n - 1] - x[n - 2];
    double a = 1.0 / hn;
    b = 2.0 / hn;
    d = 3.0 * (f[n - 1] - f[n - 2]) / (hn * hn);
    dp[n - 1] = (d - dp[n - 2] * a) / (b - cp[n - 2] * a);
    slopes[n - 1] = dp[n - 1];
    for(int i = n - 2; i >= 0; i--) slopes[i] = dp[i] - cp[i] * slopes[i + 1];
}

int geoac_grid_table_eq(int eqset, int nx, int ny, int nz, const double* x, const double* y, const double* z,
                        const double* T, const double* u, const double* v, const double* rho, double* tab){
    const bool glob = (eqset == GEOAC_EQ_GLOBAL_RNGDEP);
    if((!glob && eqset != GEOAC_EQ_3D_RNGDEP) || nx < 2 || ny < 2 || nz < 3 || !tab) return -1;
    const int nseg = nz - 1, nn = nx * ny;
    const double* F[4] = { T, u, v, rho };
    const size_t rho_off = (size_t)3 * nseg * nn * 40;
    std::vector<double> V0((size_t)nseg * nn * 4), Vx((size_t)nseg * nn * 4), Vy((size_t)nseg * nn * 4);
    std::vector<double> qx((size_t)nseg * nn, 0.0), qy((size_t)nseg * nn, 0.0);      // Q12b offsets of the Vx', Vy' rows
    std::vector<double> dcol((size_t)nz), sl((size_t)nz);
    auto at = [&](std::vector<double>& V, int k, int i, int j) -> double* { return &V[(((size_t)k * nn) + (size_t)i * ny + j) * 4]; };
    auto col_cubics = [&](const double* f, std::vector<double>& V, int i, int j, bool quirk_slopes){
        if(quirk_slopes) slopes_q12a(nz, z, f, sl.data()); else geoac_natural_spline_slopes(nz, z, f, sl.data());
        for(int k = 0; k < nseg; k++) geoac_spline_segment_cubic(z[k], z[k + 1], f[k], f[k + 1], sl[k], sl[k + 1], at(V, k, i, j), 1);
    };
    for(int f = 0; f < 4; f++){
        for(int i = 0; i < nx; i++) for(int j = 0; j < ny; j++){
            const double* col = F[f] + ((size_t)i * ny + j) * nz;
            col_cubics(col, V0, i, j, false);                       // S_f: vertical natural spline of the values
            if(f == 3) continue;                                   // rho is only ever evaluated through Eval_Spline_f
            // S_fx: spline of the centred (one-sided at the edges) difference along the first horizontal axis
            int iu = std::min(i + 1, nx - 1), id = std::max(i - 1, 0);
            const double* cu = F[f] + ((size_t)iu * ny + j) * nz; const double* cd = F[f] + ((size_t)id * ny + j) * nz;
            for(int k = 0; k < nz; k++) dcol[k] = (cu[k] - cd[k]) / (x[iu] - x[id]);
            col_cubics(dcol.data(), Vx, i, j, glob);
            if(glob) for(int k = 0; k < nseg; k++) qx[(size_t)k * nn + (size_t)i * ny + j] = (dcol[k + 1] - dcol[k]) / (z[k + 1] - z[k]);
            // S_fy
            int ju = std::min(j + 1, ny - 1), jd = std::max(j - 1, 0);
            const double* du = F[f] + ((size_t)i * ny + ju) * nz; const double* dd = F[f] + ((size_t)i * ny + jd) * nz;
            for(int k = 0; k < nz; k++) dcol[k] = (du[k] - dd[k]) / (y[ju] - y[jd]);
            col_cubics(dcol.data(), Vy, i, j, glob);
            if(glob) for(int k = 0; k < nseg; k++) qy[(size_t)k * nn + (size_t)i * ny + j] = (dcol[k + 1] - dcol[k]) / (z[k + 1] - z[k]);
        }
        for(int k = 0; k < nseg; k++) for(int i = 0; i < nx; i++) for(int j = 0; j < ny; j++){
            const int iu = std::min(i + 1, nx - 1), id = std::max(i - 1, 0), ju = std::min(j + 1, ny - 1), jd = std::max(j - 1, 0);
            const double ix = 1.0 / (x[iu] - x[id]), iy = 1.0 / (y[ju] - y[jd]);
            double* r = (f < 3) ? &tab[((((size_t)f * nseg + k) * nn) + (size_t)i * ny + j) * 40]
                                : &tab[rho_off + (((size_t)k * nn) + (size_t)i * ny + j) * 16];
            auto dx  = [&](std::vector<double>& V, int c){ return (at(V, k, iu, j)[c] - at(V, k, id, j)[c]) * ix; };
            auto dy  = [&](std::vector<double>& V, int c){ return (at(V, k, i, ju)[c] - at(V, k, i, jd)[c]) * iy; };
            auto dxy = [&](std::vector<double>& V, int c){ return (at(V, k, iu, ju)[c] - at(V, k, iu, jd)[c] - at(V, k, id, ju)[c] + at(V, k, id, jd)[c]) * (ix * iy); };
            for(int c = 0; c < 4; c++){
                r[0 + c] = at(V0, k, i, j)[c]; r[4 + c] = dx(V0, c); r[8 + c] = dy(V0, c); r[12 + c] = dxy(V0, c);
                if(f < 3){
                    r[16 + c] = at(Vx, k, i, j)[c]; r[20 + c] = dx(Vx, c); r[24 + c] = dxy(Vx, c);
                    r[28 + c] = at(Vy, k, i, j)[c]; r[32 + c] = dy(Vy, c); r[36 + c] = dxy(Vy, c);
                }
            }
            if(f < 3 && glob){
                // only the z-DERIVATIVE of the Vx / Vy cubics is ever evaluated (FX / FY rows of the df/dr patch), and the spherical
                // reference drops the difference-quotient term of that derivative ((x - x) = 0, :546, :562; Q12b): fold it into c1
                r[16 + 1] -= qx[(size_t)k * nn + (size_t)i * ny + j];
                r[28 + 1] -= qy[(size_t)k * nn + (size_t)i * ny + j];
            }
        }
    }
    return 0;
}

int geoac_grid_table(int nx, int ny, int nz, const double* x, const double* y, const double* z,
                     const double* T, const double* u, const double* v, const double* rho, double* tab){
    return geoac_grid_table_eq(GEOAC_EQ_3D_RNGDEP, nx, ny, nz, x, y, z, T, u, v, rho, tab);
}

double geoac_grid_eval(int nx, int ny, int nz, const double* x, const double* y, const double* z, const double* tab,
                       int field, double xq, double yq, double zq){
    return geoac_grid_eval_eq(GEOAC_EQ_3D_RNGDEP, nx, ny, nz, x, y, z, tab, field, xq, yq, zq);
}

double geoac_grid_eval_eq(int eqset, int nx, int ny, int nz, const double* x, const double* y, const double* z, const double* tab,
                          int field, double xq, double yq, double zq){
    const int nseg = nz - 1, nn = nx * ny;
    const bool glob = (eqset == GEOAC_EQ_GLOBAL_RNGDEP);
    xq = std::min(std::max(xq, x[0]), x[nx - 1]); yq = std::min(std::max(yq, y[0]), y[ny - 1]); zq = std::min(std::max(zq, z[0]), z[nz - 1]);
    int kx = 0, ky = 0, kz = 0;
    while(kx < nx - 2 && xq >= x[kx + 1]) kx++;
    while(ky < ny - 2 && yq >= y[ky + 1]) ky++;
    while(kz < nseg - 1 && zq > z[kz + 1]) kz++;
    const double t = zq - z[kz], dxs = x[kx + 1] - x[kx], dys = y[ky + 1] - y[ky];
    const double xs = (xq - x[kx]) / dxs, ys = (yq - y[ky]) / dys;
    auto herm = [](double s, double* h, double* g){ double s2 = s * s, s3 = s2 * s; h[1] = 3.0 * s2 - 2.0 * s3; h[0] = 1.0 - h[1]; g[0] = s - 2.0 * s2 + s3; g[1] = s3 - s2; };
    double hxh[2], hxg[2], hyh[2], hyg[2];
    herm(xs, hxh, hxg); herm(ys, hyh, hyg);
    const size_t stride = field < 3 ? 40 : 16;
    const double* base = field < 3 ? tab + (((size_t)field * nseg + kz) * nn) * 40 : tab + (size_t)3 * nseg * nn * 40 + ((size_t)kz * nn) * 16;
    auto val = [&](const double* c){ return c[0] + t * (c[1] + t * (c[2] / 2.0 + t * c[3] / 6.0)); };
    double v = 0.0;
    for(int a = 0; a < 2; a++) for(int b = 0; b < 2; b++){
        const double* r = base + ((size_t)(kx + a) * ny + (ky + b)) * stride;
        // y rows: the Cartesian scalar evaluator scales them by the x cell size (Q11); the spherical one by its own
        v += hxh[a] * hyh[b] * val(r) + hxg[a] * hyh[b] * dxs * val(r + 4) + hxh[a] * hyg[b] * (glob ? dys : dxs) * val(r + 8) + hxg[a] * hyg[b] * dxs * dys * val(r + 12);
    }
    return v;
}

void geoac_natural_spline_slopes(int n, const double* x, const double* f, double* slopes){
    // tridiagonal system of the natural spline in slope form, forward sweep + back substitution
    std::vector<double> cp(n), dp(n);
    double h0 = x[1] - x[0];
    double b = 2.0 / h0, c = 1.0 / h0, d = 3.0 * (f[1] - f[0]) / (h0 * h0);
    cp[0] = c / b; dp[0] = d / b;
    for(int i = 1; i < n - 1; i++){
        double hl = x[i] - x[i - 1], hr = x[i + 1] - x[i];
        double a = 1.0 / hl;
        b = 2.0 * (1.0 / hl + 1.0 / hr);
        c = 1.0 / hr;
        d = 3.0 * ((f[i] - f[i - 1]) / (hl * hl) + (f[i + 1] - f[i]) / (hr * hr));
        double den = b - cp[i - 1] * a;
        cp[i] = c / den;
        dp[i] = (d - dp[i - 1] * a) / den;
    }
    double hn = x[n - 1] - x[n - 2];
    double a = 1.0 / hn;
    b = 2.0 / hn;
    d = 3.0 * (f[n - 1] - f[n - 2]) / (hn * hn);
    dp[n - 1] = (d - dp[n - 2] * a) / (b - cp[n - 2] * a);
    slopes[n - 1] = dp[n - 1];
    for(int i = n - 2; i >= 0; i--) slopes[i] = dp[i] - cp[i] * slopes[i + 1];
}

long geoac_fan_enumerate(double theta_min, double theta_max, double theta_step,
                         double phi_min, double phi_max, double phi_step,
                         long cap, double* theta_out, double* phi_out){
    long n = 0;
    if(!(theta_step > 0) || !(phi_step > 0)) return -1;
    for(double phi = phi_min; phi <= phi_max; phi += phi_step){
        for(double theta = theta_min; theta <= theta_max; theta += theta_step){
            if(n < cap && theta_out && phi_out){ theta_out[n] = theta; phi_out[n] = phi; }
            n++;
        }
    }
    return n;
}

}  // extern "C"
