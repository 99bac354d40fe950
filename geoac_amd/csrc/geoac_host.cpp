// geoac_host.cpp - once-per-job host set-up (see include/geoac_host.h).  Plain C++; part of libgeoac_hip.so.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/geoac_hip.h"
#include "../../include/geoac_host.h"

namespace {
const double kREarth = 6370.0;     // G2S_GlobalSpline1D.cpp:35

bool is_global(int eqset){ return eqset == GEOAC_EQ_GLOBAL || eqset == GEOAC_EQ_GLOBAL_RNGDEP; }

// wind taper towards the ground + m/s -> km/s (G2S_Spline1D.cpp:122-124); z_grnd is 0 when the mains load the profile
double taper(double z){ return (2.0 / (1.0 + exp(-(z - 0.0) / 0.2)) - 1.0) / 1000.0; }
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
    for(int i = 0; i < n; i++){
        double xi = z[i];
        double w;
        if(is_global(eqset)){
            xi += kREarth;                           // G2S_GlobalSpline1D.cpp:128
            w = taper(xi - kREarth);                 // :129 evaluates (r - r_earth - z_grnd)
        } else {
            w = taper(xi);
        }
        x[i] = xi; T_out[i] = T[i]; rho[i] = rho_in[i];
        u[i] = u_ms[i] * w;
        v[i] = v_ms[i] * w;
    }
    return n;
}

int geoac_met_load(const char* file, const char* format, int eqset, int cap,
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
    return geoac_met_from_columns(eqset, rows, z.data(), Tc.data(), uc.data(), vc.data(), rc.data(), x, T, u, v, rho);
}

int geoac_grid_dims(const char* prefix, const char* locx, const char* locy, int* nx, int* ny, int* nz){
    std::string first = std::string(prefix) + "0.met";
    *nx = geoac_met_rows(locx); *ny = geoac_met_rows(locy); *nz = geoac_met_rows(first.c_str());
    return (*nx >= 2 && *ny >= 2 && *nz >= 3) ? 0 : -1;
}

int geoac_grid_load(const char* prefix, const char* locx, const char* locy, const char* format, double z_grnd,
                    int nx, int ny, int nz, double* x, double* y, double* z,
                    double* T, double* u, double* v, double* rho){
    int fmt;
    if(strncmp(format, "zTuvdp", 6) == 0) fmt = 0;
    else if(strncmp(format, "zuvwTdp", 7) == 0) fmt = 1;
    else return -2;
    FILE* fp = fopen(locx, "r"); if(!fp) return -1;
    for(int i = 0; i < nx; i++) if(fscanf(fp, "%lf", &x[i]) != 1) x[i] = 0.0;
    fclose(fp);
    fp = fopen(locy, "r"); if(!fp) return -1;
    for(int j = 0; j < ny; j++) if(fscanf(fp, "%lf", &y[j]) != 1) y[j] = 0.0;
    fclose(fp);
    const int ncol = fmt ? 7 : 6;
    for(int i = 0; i < nx; i++) for(int j = 0; j < ny; j++){
        std::string name = std::string(prefix) + std::to_string(i * ny + j) + ".met";
        fp = fopen(name.c_str(), "r");
        if(!fp) return -1;
        for(int k = 0; k < nz; k++){
            double t[7] = {0, 0, 0, 0, 0, 0, 0};
            for(int q = 0; q < ncol; q++) if(fscanf(fp, "%lf", &t[q]) != 1) t[q] = 0.0;
            double zz, TT, uu, vv, rr;
            if(fmt == 0){ zz = t[0]; TT = t[1]; uu = t[2]; vv = t[3]; rr = t[4]; }
            else        { zz = t[0]; uu = t[1]; vv = t[2]; TT = t[4]; rr = t[5]; }
            z[k] = zz;
            double w = (2.0 / (1.0 + exp(-(zz - z_grnd) / 0.05)) - 1.0) / 1000.0;      // G2S_MultiDimSpline3D.cpp:167-168
            size_t o = ((size_t)i * ny + j) * nz + k;
            T[o] = TT; u[o] = uu * w; v[o] = vv * w; rho[o] = rr;
        }
        fclose(fp);
    }
    return 0;
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
