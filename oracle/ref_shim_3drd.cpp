/* TEST INFRASTRUCTURE ONLY.  Shim around the UNMODIFIED reference GeoAc3D.RngDep translation units
 * (compiled from /root/reference by oracle/Makefile, never copied).  It calls the reference's own
 * functions in the order GeoAc3D_RngDep_RunProp does (GeoAc3D.RngDep_main.cpp:244-328) and returns
 * binary-double records instead of 6-8 digit text.  See oracle/ref_shim.h for the ABI.
 */
#include <math.h>
#include <string.h>
#include <algorithm>

#include "GeoAc.Parameters.h"
#include "Atmo_State.h"
#include "G2S_MultiDimSpline3D.h"
#include "GeoAc.EquationSets.h"
#include "GeoAc.Solver.h"
#include "GeoAc.Interface.h"

#include "ref_shim.h"

/* the header copy under Code/GeoAc declares a 3-argument form; the definition (Code/Atmo/G2S_MultiDimSpline3D.cpp:1603) takes 4 */
void Spline_Multi_G2S(char*, char*, char*, char*);

static double** g_solution = 0;
static int      g_length = 0;
static int      g_eqcnt_built = 0;

static void ensure_solution(){
    double ds = 0.1;
    int length = GeoAc_ray_limit*int(1.0/ds);                       /* GeoAc3D.RngDep_main.cpp:185-186: 50 000 rows */
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

extern "C" int ref_load_grid(const char* prefix, const char* locx, const char* locy, const char* format, double z_grnd_at_load){
    z_grnd = z_grnd_at_load;                                        /* parsed BEFORE the load in this main (:135 vs :168) */
    tweak_abs = 0.3;
    Spline_Multi_G2S((char*)prefix, (char*)locx, (char*)locy, (char*)format);
    GeoAc_SetPropRegion();                                          /* :169 */
    return Temp_Spline.length_z;
}

static void apply_cfg(const ref_fan_cfg* cfg){
    z_grnd = cfg->z_grnd;
    tweak_abs = cfg->tweak_abs;
    if(cfg->vert_limit == cfg->vert_limit)   GeoAc_vert_limit  = cfg->vert_limit;
    if(cfg->xy_limits[0] == cfg->xy_limits[0]) GeoAc_x_min_limit = cfg->xy_limits[0];
    if(cfg->xy_limits[1] == cfg->xy_limits[1]) GeoAc_x_max_limit = cfg->xy_limits[1];
    if(cfg->xy_limits[2] == cfg->xy_limits[2]) GeoAc_y_min_limit = cfg->xy_limits[2];
    if(cfg->xy_limits[3] == cfg->xy_limits[3]) GeoAc_y_max_limit = cfg->xy_limits[3];
    bool CalcAmp = cfg->calc_amp != 0;
    if(cfg->mode & GEOAC_MODE_WRITE_CAUSTICS) CalcAmp = true;      /* GeoAc3D.RngDep_main.cpp:166 */
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
    double x_src = cfg->src[0], y_src = cfg->src[1], z_src = cfg->src[2];
    const double freq = cfg->freq;
    z_src = std::max(z_grnd, z_src);                                /* GeoAc3D.RngDep_main.cpp:165 */

    memset(rec, 0, sizeof(double) * (size_t)n * (bounces + 1) * GEOAC_REC_STRIDE);
    int64_t total_steps = 0, ns = 0;
    double D = 0, D_prev = 0, travel_time_sum, attenuation, z_max;
    int k = 0; bool BreakCheck;

    for(int i = 0; i < n; i++){
        double theta = theta_deg[i], phi = phi_deg[i];
        GeoAc_theta = theta*Pi/180.0;                               /* GeoAc3D.RngDep_main.cpp:247 */
        GeoAc_phi = Pi/2.0 - phi*Pi/180.0;                          /* :229 */
        GeoAc_SetInitialConditions(solution, x_src, y_src, z_src);
        travel_time_sum = 0.0; attenuation = 0.0; z_max = 0.0;

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
                            S[4] = solution[m][0];
                            S[5] = solution[m][1];
                            S[6] = std::max(solution[m][2],0.0);
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
                            S[4] = solution[m][0];
                            S[5] = solution[m][1];
                            S[6] = solution[m][2];
                            S[7] = 0.0;
                            S[8] = travel_time_sum; S[9] = 0;
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
            z_max = 0.0;                                                    /* :296: per leg */
            for(int m = 0; m < k; m++) z_max = std::max(z_max, solution[m][2]);

            double inclination = - asin(c(solution[k][0], solution[k][1], z_grnd) / c(x_src, y_src, z_src) * solution[k][5]) * 180.0 / Pi;
            double back_az = 90.0 - atan2(-solution[k][4], -solution[k][3]) * 180.0 / Pi;
            while(back_az < -180.0) back_az +=360.0;
            while(back_az >  180.0) back_az -=360.0;

            R[GEOAC_REC_VALID]  = 1.0;
            R[GEOAC_REC_TURN]   = z_max;
            R[GEOAC_REC_INCL]   = inclination;
            R[GEOAC_REC_BACKAZ] = back_az;
            R[GEOAC_REC_RANGE]  = sqrt(solution[k][0]*solution[k][0] + solution[k][1]*solution[k][1]);
            if(CalcAmp){
                R[GEOAC_REC_AMP]   = GeoAc_Amplitude(solution,k);
                R[GEOAC_REC_JACOB] = GeoAc_Jacobian(solution,k);
            }
            for(int e = 0; e < GeoAc_EqCnt; e++) R[GEOAC_REC_STATE + e] = solution[k][e];

            GeoAc_SetReflectionConditions(solution,k);
        }
        GeoAc_ClearSolutionArray(solution,k);                       /* GeoAc3D.RngDep_main.cpp:322 */
    }
    if(n_smp) *n_smp = ns;
    return total_steps;
}

extern "C" void ref_atmo_probe(int, const double*, double*, double*){}
extern "C" void ref_absorption_probe(int, const double*, const double*, double, double, double*){}
extern "C" int ref_tables(int, double*, double*, double*, double*, double*, double*, double*, double*, double*){ return -1; }

extern "C" void ref_grid_probe(int n, const double* x, const double* y, const double* z, double* out30, double* api8){
    for(int i = 0; i < n; i++){
        struct MultiDimSpline_3D* S[3] = { &Temp_Spline, &Windu_Spline, &Windv_Spline };
        for(int f = 0; f < 3; f++){
            double* o = out30 + 30*i + 10*f;
            Eval_Spline_AllOrder2(x[i], y[i], z[i], *S[f], o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9]);
        }
        double* a = api8 + 8*i;
        a[0] = c(x[i], y[i], z[i]); a[1] = rho(x[i], y[i], z[i]); a[2] = u(x[i], y[i], z[i]); a[3] = v(x[i], y[i], z[i]);
        a[4] = c_diff(x[i], y[i], z[i], 2); a[5] = u_diff(x[i], y[i], z[i], 2); a[6] = v_diff(x[i], y[i], z[i], 2);
        a[7] = c_diff(x[i], y[i], z[i], 0);
    }
}

extern "C" int ref_trace_leg0(const ref_fan_cfg* cfg, double theta_deg, double phi_deg, int max_rows, double* out, int* E){
    apply_cfg(cfg);
    double z_src = std::max(z_grnd, cfg->src[2]);
    GeoAc_theta = theta_deg*Pi/180.0;
    GeoAc_phi = Pi/2.0 - phi_deg*Pi/180.0;
    GeoAc_SetInitialConditions(g_solution, cfg->src[0], cfg->src[1], z_src);
    bool BreakCheck;
    int k = GeoAc_Propagate_RK4(g_solution, BreakCheck);
    *E = GeoAc_EqCnt;
    for(int m = 0; m <= k && m < max_rows; m++)
        for(int e = 0; e < GeoAc_EqCnt; e++) out[(size_t)m*GeoAc_EqCnt + e] = g_solution[m][e];
    GeoAc_ClearSolutionArray(g_solution, k);
    return BreakCheck ? -k : k;
}
