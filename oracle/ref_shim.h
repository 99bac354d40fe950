/* TEST INFRASTRUCTURE ONLY - never linked or loaded by the product path.
 *
 * Common C ABI of the "reference shims": oracle/ref_shim_<set>.cpp are compiled TOGETHER WITH the
 * unmodified reference translation units (from /root/reference, see oracle/Makefile) into
 * oracle/_ref/libref_<set>.so.  A shim only drives the reference's own functions in the order
 * the reference's *_RunProp loops do, and hands back full-precision (binary double) records,
 * because the reference's .dat files carry 6-8 significant digits only.
 *
 * The record layout is the same one the product's C ABI (include/geoac_hip.h) and the CPU
 * restatement (oracle/geoac_oracle.h) use, so the three can be compared field by field.
 */
#ifndef GEOAC_REF_SHIM_H_
#define GEOAC_REF_SHIM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* record / sample layouts and mode bits are the product ABI's */
#include "../include/geoac_hip.h"

typedef struct {
    double z_grnd;        /* set AFTER the profile load, like the reference's parse order          */
    double tweak_abs;
    double freq;
    double vert_limit;    /* NaN = keep what GeoAc_SetPropRegion chose                             */
    double range_limit;   /* NaN = keep                                                            */
    double src[3];        /* Global: z_src[km], lat_src[deg], lon_src[deg]; 3D: x,y,z; 2D: r(unused)=0, z_src, azimuth handled by caller */
    int    bounces;
    int    calc_amp;
    int    mode;
    int    pad_;
    double xy_limits[4];  /* RngDep sets: x_min, x_max, y_min, y_max limits; NaN = keep what GeoAc_SetPropRegion chose */
} ref_fan_cfg;

/* Spline_Single_G2S(file, format); returns number of profile rows */
int     ref_load(const char* met_file, const char* format);

/* run rays (theta_deg[i], phi_deg[i]), i < n.  rec: n*(bounces+1)*GEOAC_REC_STRIDE doubles (zeroed by
 * the callee).  smp: room for smp_cap sample records (may be NULL/0); *n_smp receives the count.
 * returns total steps = sum of GeoAc_Propagate_RK4 return values. */
int64_t ref_fan(const ref_fan_cfg* cfg, int n, const double* theta_deg, const double* phi_deg,
                double* rec, double* smp, int64_t smp_cap, int64_t* n_smp);

/* atmosphere probes: out[9*i+..] = c, c', c'', u, u', u'', v, v', v'' and rho_out[i] at coordinate x[i]
 * (Global: geocentric radius; Cartesian: altitude) */
void    ref_atmo_probe(int n, const double* x, double* out9, double* rho_out);
/* SuthBass_Alpha at coordinate x[i], frequency f[i] */
void    ref_absorption_probe(int n, const double* x, const double* f, double z_grnd, double tweak, double* out);
/* spline tables as the reference built them: fills x,T,u,v,rho and the four slope arrays (each n doubles) */
int     ref_tables(int cap, double* x, double* T, double* u, double* v, double* rho,
                   double* sT, double* su, double* sv, double* srho);
/* range-dependent sets only: Spline_Multi_G2S(prefix, locx, locy, format) after setting z_grnd (the RngDep mains parse
 * z_grnd= before loading); and probes of the grid interpolant (layout as orc_grid_probe) */
int     ref_load_grid(const char* prefix, const char* locx, const char* locy, const char* format, double z_grnd_at_load);
void    ref_grid_probe(int n, const double* x, const double* y, const double* z, double* out30, double* api8);
/* full state row dump of one leg-0 propagation (for stepper-level tests): out rows*(E) doubles, returns k */
int     ref_trace_leg0(const ref_fan_cfg* cfg, double theta_deg, double phi_deg, int max_rows, double* out, int* E);

#ifdef __cplusplus
}
#endif
#endif
