/* TEST INFRASTRUCTURE ONLY.  Shim around the UNMODIFIED reference GeoAcGlobal.RngDep translation units
 * (compiled from /root/reference by oracle/Makefile, never copied).  It calls the reference's own
 * functions in the order GeoAcGlobal_RngDep_RunProp does (GeoAcGlobal.RngDep_main.cpp:251-337) and returns
 * binary-double records instead of 6-8 digit text.  See oracle/ref_shim.h for the ABI.
 */
#include <math.h>
#include <string.h>
#include <algorithm>

#include "GeoAc.Parameters.h"
#include "Atmo_State.h"
#include "G2S_GlobalMultiDimSpline3D.h"
#include "GeoAc.EquationSets.h"
#include "GeoAc.Solver.h"
#include "GeoAc.Interface.h"

#include "ref_shim.h"

/* the header copy under Code/GeoAc declares a 3-argument form; the definition (Code/Atmo/G2S_GlobalMultiDimSpline3D.cpp:1473) takes 4 */
void Spline_Multi_G2S(char*, char*, char*, char*);

static double** g_solution = 0;
static int      g_length = 0;
static int      g_eqcnt_built = 0;

static void ensure_solution(){
    int length = GeoAc_ray_limit*int(1.0/(GeoAc_ds_min*10));       /* GeoAcGlobal.RngDep_main.cpp: same expression as the stratified main */
    if(g_solution && (g_length != length || g_eqcnt_built != GeoAc_EqCnt)){
        GeoAc_DeleteSolutionArray(g_solution, g_length);
        g_solution = 0;
    }
    if(!g_solution){
        GeoAc_BuildSolutionArray(g_solution, length);
        g_length = length; g_eqcnt_built = GeoAc_EqCnt;
    }
}

extern "C" int ref_load(const char*, const char*){ return -1; }       /* 1-D loader: not applicable to this set */

/* loc files hold latitudes / longitudes in degrees.  -prop never parses z_grnd= (the load sees the global's current value);
 * -interactive / -eig_* parse it before loading (:350-355): z_grnd_at_load covers both */
extern "C" int ref_load_grid(const char* prefix, const char* loclat, const char* loclon, const char* format, double z_grnd_at_load){
    z_grnd = z_grnd_at_load;
    tweak_abs = 0.3;
    Spline_Multi_G2S((char*)prefix, (char*)loclat, (char*)loclon, (char*)format);     /* calls GeoAc_SetPropRegion itself (:1484) */
    return Temp_Spline.length_r;
}

/* defaults of the source position: centre of the grid, degrees (GeoAcGlobal.RngDep_main.cpp:135-137) */
extern "C" void ref_grid_centre(double* lat_deg, double* lon_deg){
    *lat_deg = (t_vals[0] + t_vals[t_cnt-1])/2.0 * 180.0/Pi;
    *lon_deg = (p_vals[0] + p_vals[p_cnt-1])/2.0 * 180.0/Pi;
}

static void apply_cfg(const ref_fan_cfg* cfg){
    z_grnd = cfg->z_grnd;
    tweak_abs = cfg->tweak_abs;
    if(cfg->vert_limit == cfg->vert_limit)   GeoAc_vert_limit  = cfg->vert_limit;
    /* lat_min= / lat_max= / lon_min= / lon_max= are assigned raw (:166-169) and compared with radians (GlobalRngDep.cpp:530-531) */
    if(cfg->xy_limits[0] == cfg->xy_limits[0]) GeoAc_lat_min_limit = cfg->xy_limits[0];
    if(cfg->xy_limits[1] == cfg->xy_limits[1]) GeoAc_lat_max_limit = cfg->xy_limits[1];
    if(cfg->xy_limits[2] == cfg->xy_limits[2]) GeoAc_lon_min_limit = cfg->xy_limits[2];
    if(cfg->xy_limits[3] == cfg->xy_limits[3]) GeoAc_lon_max_limit = cfg->xy_limits[3];
    bool CalcAmp = cfg->calc_amp != 0;
    if(cfg->mode & GEOAC_MODE_WRITE_CAUSTICS) CalcAmp = true;      /* :178 */
    GeoAc_ConfigureCalcAmp(CalcAmp);
    ensure_solution();
}

extern "C" int64_t ref_fan(const ref_fan_cfg* cfg, int n, const double* theta_deg, const double* phi_deg,
                           double* rec, double* smp, int64_t smp_cap, int64_t* n_smp){
    apply_cfg(cfg);
    double** solution = g_solution;
    const bool CalcAmp = GeoAc_CalcAmp;
    const bool WriteRays = (cfg->mode & GEOAC_MODE_WRITE_RAYS) != 0;
    const bool WriteCaustics = (cfg->mode & GEOAC_MODE_WRITE_CAUSTICS) != 0;
    const int bounces = cfg->bounces;
    double z_src = cfg->src[0], lat_src = cfg->src[1], lon_src = cfg->src[2];
    const double freq = cfg->freq;
    z_src = std::max(z_src, z_grnd);                                /* :177 */

    memset(rec, 0, sizeof(double) * (size_t)n * (bounces + 1) * GEOAC_REC_STRIDE);
    int64_t total_steps = 0, ns = 0;
    double D = 0, D_prev = 0, travel_time_sum, attenuation, r_max;
    int k = 0; bool BreakCheck;

    for(int i = 0; i < n; i++){
        double theta = theta_deg[i], phi = phi_deg[i];
        GeoAc_theta = theta*Pi/180.0;                               /* :254 */
        GeoAc_phi = Pi/2.0 - phi*Pi/180.0;                          /* :255 */
        GeoAc_SetInitialConditions(solution, z_src, lat_src*Pi/180.0, lon_src*Pi/180.0);
        travel_time_sum = 0.0; attenuation = 0.0;

        for(int bnc_cnt = 0; bnc_cnt <= bounces; bnc_cnt++){
            double* R = rec + ((size_t)i * (bounces + 1) + bnc_cnt) * GEOAC_REC_STRIDE;
            k = GeoAc_Propagate_RK4(solution, BreakCheck);
            total_steps += k;
            R[GEOAC_REC_STEPS] = k;
            R[GEOAC_REC_BROKE] = BreakCheck ? 1.0 : 0.0;

            if(WriteRays || WriteCaustics){
                if(WriteCaustics) D_prev = GeoAc_Jacobian(solution,1);
                for(int m = 1; m < k; m++){
                    if(WriteCaustics) D = GeoAc_Jacobian(solution,m);
                    GeoAc_TravelTimeSegment(travel_time_sum, solution, m-1, m);
                    GeoAc_SB_AttenSegment(attenuation, solution, m-1, m, freq);
                    if(WriteRays && m % 25 == 0){
                        if(smp && ns < smp_cap){
                            double* S = smp + ns * GEOAC_SMP_STRIDE;
                            S[GEOAC_SMP_RAY] = i; S[GEOAC_SMP_LEG] = bnc_cnt; S[GEOAC_SMP_M] = m; S[GEOAC_SMP_KIND] = 0;
                            S[4] = solution[m][0] - r_earth;
                            S[5] = solution[m][1] * 180.0/Pi;
                            S[6] = solution[m][2] * 180.0/Pi;
                            S[7] = CalcAmp ? 20.0*log10(GeoAc_Amplitude(solution,m)) : 0.0;
                            S[8] = -attenuation;
                            S[9] = travel_time_sum;
                        }
                        ns++;
                    }
                    if(WriteCaustics && D*D_prev < 0.0){
                        if(smp && ns < smp_cap){
                            double* S = smp + ns * GEOAC_SMP_STRIDE;
                            S[GEOAC_SMP_RAY] = i; S[GEOAC_SMP_LEG] = bnc_cnt; S[GEOAC_SMP_M] = m; S[GEOAC_SMP_KIND] = 1;
                            S[4] = solution[m][0] - r_earth;
                            S[5] = solution[m][1] * 180.0/Pi;
                            S[6] = solution[m][2] * 180.0/Pi;
                            S[7] = travel_time_sum;
                            S[8] = 0; S[9] = 0;
                        }
                        ns++;
                    }
                    if(WriteCaustics) D_prev = D;
                }
            } else {
                travel_time_sum += GeoAc_TravelTime(solution, k);
                attenuation += GeoAc_SB_Atten(solution, k, freq);
            }
            R[GEOAC_REC_TTIME] = travel_time_sum;
            R[GEOAC_REC_ATTEN] = attenuation;

            if(BreakCheck) break;
            r_max = 0.0;                                                    /* :304: per leg (Q8) */
            for(int m = 0; m < k; m++) r_max = std::max(r_max, solution[m][0] - r_earth);

            double GC_Dist1 = pow(sin((solution[k][1] - lat_src*Pi/180.0)/2.0),2);
            double GC_Dist2 = cos(lat_src*Pi/180.0) * cos(solution[k][1]) * pow(sin((solution[k][2] - lon_src*Pi/180.0)/2.0),2);
            /* no leading minus in this main (Q10, :310) */
            double inclination = asin(c(solution[k][0], solution[k][1], solution[k][2]) / c(r_earth + z_src, lat_src*Pi/180.0, lon_src*Pi/180.0) * solution[k][3]) * 180.0 / Pi;
            double back_az = 90.0 - atan2(-solution[k][4], -solution[k][5]) * 180.0 / Pi;
            if(back_az < -180.0) back_az += 360.0;
            if(back_az >  180.0) back_az -= 360.0;

            R[GEOAC_REC_VALID]  = 1.0;
            R[GEOAC_REC_TURN]   = r_max;
            R[GEOAC_REC_INCL]   = inclination;
            R[GEOAC_REC_BACKAZ] = back_az;
            R[GEOAC_REC_RANGE]  = 2.0 * r_earth * asin(sqrt(GC_Dist1+GC_Dist2));
            if(CalcAmp){
                R[GEOAC_REC_AMP]   = GeoAc_Amplitude(solution,k);
                R[GEOAC_REC_JACOB] = GeoAc_Jacobian(solution,k);
            }
            for(int e = 0; e < GeoAc_EqCnt; e++) R[GEOAC_REC_STATE + e] = solution[k][e];

            GeoAc_SetReflectionConditions(solution,k);
        }
        GeoAc_ClearSolutionArray(solution,k);                       /* :333 */
    }
    if(n_smp) *n_smp = ns;
    return total_steps;
}

extern "C" void ref_atmo_probe(int, const double*, double*, double*){}
extern "C" void ref_absorption_probe(int, const double*, const double*, double, double, double*){}
extern "C" int ref_tables(int, double*, double*, double*, double*, double*, double*, double*, double*, double*){ return -1; }

/* here (x, y, z) = (r, lat, lon); api8 = c, rho, u, v, dc/dr, du/dr, dv/dr, dc/dlat */
extern "C" void ref_grid_probe(int n, const double* x, const double* y, const double* z, double* out30, double* api8){
    for(int i = 0; i < n; i++){
        struct MultiDimSpline_3DGlobal* S[3] = { &Temp_Spline, &Windu_Spline, &Windv_Spline };
        for(int f = 0; f < 3; f++){
            double* o = out30 + 30*i + 10*f;
            Eval_Spline_AllOrder2(x[i], y[i], z[i], *S[f], o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9]);
        }
        double* a = api8 + 8*i;
        a[0] = c(x[i], y[i], z[i]); a[1] = rho(x[i], y[i], z[i]); a[2] = u(x[i], y[i], z[i]); a[3] = v(x[i], y[i], z[i]);
        a[4] = c_diff(x[i], y[i], z[i], 0); a[5] = u_diff(x[i], y[i], z[i], 0); a[6] = v_diff(x[i], y[i], z[i], 0);
        a[7] = c_diff(x[i], y[i], z[i], 1);
    }
}

extern "C" int ref_trace_leg0(const ref_fan_cfg* cfg, double theta_deg, double phi_deg, int max_rows, double* out, int* E){
    apply_cfg(cfg);
    double z_src = std::max(cfg->src[0], z_grnd);
    GeoAc_theta = theta_deg*Pi/180.0;
    GeoAc_phi = Pi/2.0 - phi_deg*Pi/180.0;
    GeoAc_SetInitialConditions(g_solution, z_src, cfg->src[1]*Pi/180.0, cfg->src[2]*Pi/180.0);
    bool BreakCheck;
    int k = GeoAc_Propagate_RK4(g_solution, BreakCheck);
    *E = GeoAc_EqCnt;
    for(int m = 0; m <= k && m < max_rows; m++)
        for(int e = 0; e < GeoAc_EqCnt; e++) out[(size_t)m*GeoAc_EqCnt + e] = g_solution[m][e];
    GeoAc_ClearSolutionArray(g_solution, k);
    return BreakCheck ? -k : k;
}
