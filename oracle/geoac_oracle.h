/* TEST INFRASTRUCTURE ONLY - the product path (geoac_amd/, libgeoac_hip.so) never includes, links,
 * loads or executes anything in oracle/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker.
 *
 * geoac_oracle: plain-C CPU restatement of GeoAc's ray-fan hot path (RK4 stepper, the 2D / 3D /
 * Global stratified equation sets, the 1-D cubic-spline atmosphere, Sutherland-Bass absorption and
 * the *_RunProp fan / bounce / post-pass loops).  Every function cites the reference file:line it
 * restates.  Parity pin: tests/test_oracle_vs_ref.py checks it BIT-FOR-BIT against the compiled
 * reference (oracle/_ref/libref_*.so) where /root/reference is available, and against the golden
 * vectors in tests/golden/ (generated from that compiled reference by tests/golden/make_golden.py)
 * everywhere else.
 */
#ifndef GEOAC_ORACLE_H_
#define GEOAC_ORACLE_H_

#include <stdint.h>
#include "ref_shim.h"      /* ref_fan_cfg + record layout (same ABI as the reference shims) */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

orc_ctx* orc_create(int eqset);                 /* GEOAC_EQ_2D / _3D / _GLOBAL */
void     orc_destroy(orc_ctx*);

/* read a .met profile like Spline_Single_G2S does; returns row count (<0 on error) */
int      orc_load(orc_ctx*, const char* met_file, const char* format);
/* same from arrays holding the file's raw columns z[km], T[K], u,v[m/s], rho (taper + km/s conversion applied here) */
int      orc_load_arrays(orc_ctx*, int n, const double* z, const double* T, const double* u,
                         const double* v, const double* rho);

int64_t  orc_fan(orc_ctx*, const ref_fan_cfg* cfg, int n, const double* theta_deg, const double* phi_deg,
                 double* rec, double* smp, int64_t smp_cap, int64_t* n_smp);

void     orc_atmo_probe(orc_ctx*, int n, const double* x, double* out9, double* rho_out);
void     orc_absorption_probe(orc_ctx*, int n, const double* x, const double* f, double z_grnd, double tweak, double* out);
int      orc_tables(orc_ctx*, int cap, double* x, double* T, double* u, double* v, double* rho,
                    double* sT, double* su, double* sv, double* srho);
int      orc_trace_leg0(orc_ctx*, const ref_fan_cfg* cfg, double theta_deg, double phi_deg, int max_rows, double* out, int* E);

/* range-dependent Cartesian set (GEOAC_EQ_3D_RNGDEP): grid of profiles <prefix><n>.met, n = ix*ny + iy */
int      orc_load_grid(orc_ctx*, const char* prefix, const char* locx, const char* locy, const char* format, double z_grnd_at_load);
void     orc_grid_dims(orc_ctx*, int* nx, int* ny, int* nz);
/* range-dependent spherical set (GEOAC_EQ_GLOBAL_RNGDEP): same entry points; locx / locy = latitude / longitude node files
 * [deg], probes take (r, lat, lon) [km, rad, rad].  orc_grid_centre: the mains' default source position [deg] */
void     orc_grid_centre(orc_ctx*, double* lat_deg, double* lon_deg);
void     orc_grid_probe(orc_ctx*, int n, const double* x, const double* y, const double* z, double* out30, double* api8);

/* limits chosen by GeoAc_SetPropRegion for the loaded profile */
void     orc_limits(orc_ctx*, double* vert_limit, double* range_limit);

#ifdef __cplusplus
}
#endif
#endif
