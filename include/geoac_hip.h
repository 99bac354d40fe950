/* geoac_hip.h - C ABI of the MI355X ray-fan integrator (libgeoac_hip.so).
 *
 * This is the drop-in boundary for GeoAc's hot path.  The reference (LANL-Seismoacoustics/GeoAc)
 * has no FFI layer; what this library replaces is the internal free-function seam that the five
 * `*_RunProp` drivers and the eigenray search call:
 *
 *   int  GeoAc_Propagate_RK4(double**& solution, bool& check)        Code/GeoAc/GeoAc.Solver.h:8
 *   the 20 functions of                                              Code/GeoAc/GeoAc.EquationSets.h:6-31
 *   c,u,v,w,rho and X_diff / X_ddiff, SuthBass_Alpha                   Code/Atmo/Atmo_State.h:17-36
 *   the launch-angle double loop + bounce loop + post-pass           Code/GeoAcGlobal_main.cpp:241-325,
 *                                                                    Code/GeoAc3D_main.cpp:226-307,
 *                                                                    Code/GeoAc2D_main.cpp:170-232
 *
 * The reference passes inputs through process-wide globals (GeoAc.Parameters.h:7-39) and keeps the
 * whole ray in a caller-owned `double** solution`; here inputs travel in `geoac_params`, one call
 * integrates a whole fan of launch angles on the GPU, and what comes back is one fixed-stride record
 * per (ray, bounce leg) holding everything a `_results.dat` row is made of, plus (optionally) the
 * decimated `_raypaths.dat` samples.
 *
 * Conventions: plain C, no exceptions cross the boundary, every call returns 0 on success or a
 * negative GEOAC_E* code (geoac_strerror).  Pointers are host pointers unless the name says `dev`.
 * Inputs are borrowed for the duration of the call.  One fan may be in flight per context; several
 * contexts (one per GPU / per process) may run concurrently.  The library never falls back to a CPU
 * path: without a usable HIP device every compute entry point fails with GEOAC_E_NODEVICE.
 */
#ifndef GEOAC_HIP_H_
#define GEOAC_HIP_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- equation sets (one reference executable each, makefile:10-23) ---- */
enum {
    GEOAC_EQ_2D            = 0,   /* GeoAc.EquationSets.2DStratified.cpp   E = 3 / 6   */
    GEOAC_EQ_3D            = 1,   /* GeoAc.EquationSets.3DStratified.cpp   E = 4 / 12  */
    GEOAC_EQ_GLOBAL        = 2,   /* GeoAc.EquationSets.Global.cpp         E = 6 / 18  */
    GEOAC_EQ_3D_RNGDEP     = 3,   /* GeoAc.EquationSets.3DRngDep.cpp       E = 6 / 18  */
    GEOAC_EQ_GLOBAL_RNGDEP = 4    /* GeoAc.EquationSets.GlobalRngDep.cpp */
};

/* ---- status codes ---- */
enum {
    GEOAC_OK            =  0,
    GEOAC_E_INVALID     = -1,   /* bad argument / call order                       */
    GEOAC_E_NODEVICE    = -2,   /* no usable HIP device (no CPU fallback exists)   */
    GEOAC_E_HIP         = -3,   /* a HIP runtime call failed (see geoac_last_error) */
    GEOAC_E_UNSUPPORTED = -4,   /* equation set / mode not implemented             */
    GEOAC_E_CAPACITY    = -5,   /* a ray exceeded step_limit or a buffer was too small */
    GEOAC_E_NOMEM       = -6
};

/* ---- one record per (ray, leg); leg = bounce index 0..bounces ---- */
#define GEOAC_REC_STRIDE 32
enum {
    GEOAC_REC_VALID   = 0,   /* 1.0: the reference would write a results row for this leg        */
    GEOAC_REC_STEPS   = 1,   /* return value of GeoAc_Propagate_RK4 for this leg (0 if not run)  */
    GEOAC_REC_BROKE   = 2,   /* 1.0: BreakCheck ended this leg (ray abandoned, no row)           */
    GEOAC_REC_TTIME   = 3,   /* travel_time_sum, cumulative over legs [s]                         */
    GEOAC_REC_ATTEN   = 4,   /* attenuation, cumulative, positive [dB]                            */
    GEOAC_REC_TURN    = 5,   /* turning height r_max / z_max [km]                                 */
    GEOAC_REC_INCL    = 6,   /* arrival inclination [deg]                                         */
    GEOAC_REC_BACKAZ  = 7,   /* back azimuth [deg]                                                */
    GEOAC_REC_AMP     = 8,   /* GeoAc_Amplitude(solution,k), linear (row prints 20 log10); 0 if CalcAmp off */
    GEOAC_REC_RANGE   = 9,   /* Global: great-circle range [km] (celerity = range / ttime); 3D: sqrt(x^2+y^2); 2D: r */
    GEOAC_REC_JACOB   = 10,  /* GeoAc_Jacobian(solution,k); 0 if CalcAmp off                      */
    GEOAC_REC_STATE   = 12   /* solution[k][0..E-1]  (E <= 18), the first sub-ground sample (Q2)  */
};

/* ---- one sample record per `_raypaths.dat` row (every 25th step) or caustic row ---- */
#define GEOAC_SMP_STRIDE 10
enum {
    GEOAC_SMP_RAY  = 0, GEOAC_SMP_LEG = 1, GEOAC_SMP_M = 2,
    GEOAC_SMP_KIND = 3,      /* 0 = raypath row, 1 = caustic row                                  */
    GEOAC_SMP_V0   = 4       /* the row's columns, unformatted                                    */
};

/* ---- mode bits ---- */
#define GEOAC_MODE_WRITE_RAYS      1   /* WriteRays=True: post-pass sums segments 0..k-2 (GeoAcGlobal_main.cpp:264-267), samples kept */
#define GEOAC_MODE_WRITE_CAUSTICS  2   /* WriteCaustics=True (forces CalcAmp, GeoAcGlobal_main.cpp:166) */
#define GEOAC_MODE_INTERACTIVE     4   /* bookkeeping of the -interactive loops: GeoAc2D's "turning height" there is the running maximum of
                                          solution[m][2] = nu_z over rows 1..k-1 (GeoAc2D_main.cpp:332), reported in GEOAC_REC_TURN */

/* ---- parameters: the reference's globals (GeoAc.Parameters*.cpp) and *_RunProp locals ---- */
typedef struct {
    double ds_min;        /* GeoAc_ds_min   0.001                                   */
    double ds_max;        /* GeoAc_ds_max   0.5                                     */
    double ray_limit;     /* GeoAc_ray_limit 5000 (Cartesian) / 10000 (Global)      */
    double vert_limit;    /* GeoAc_vert_limit  (Global: geocentric radius)          */
    double range_limit;   /* GeoAc_range_limit                                      */
    double z_grnd;        /* ground elevation [km]                                  */
    double r_earth;       /* 6370.0 (Global only)                                   */
    double tweak_abs;     /* abs_coeff, 0.3                                         */
    double freq;          /* Hz, 0.1                                                */
    double src[3];        /* Global: z_src [km], lat_src [deg], lon_src [deg]; 3D: x,y,z [km]; 2D: z_src, -, - */
    int    bounces;       /* legs = bounces + 1                                     */
    int    calc_amp;      /* CalcAmp                                                */
    int    mode;          /* GEOAC_MODE_* bits                                      */
    int    sample_stride; /* 25 (GeoAcGlobal_main.cpp:269)                          */
    double xy_limits[4];  /* RngDep sets: GeoAc_x_min/x_max/y_min/y_max_limit (3D) or lat_min/lat_max/lon_min/lon_max [rad] (Global); NaN = grid extents (GeoAc_SetPropRegion) */
} geoac_params;

typedef struct geoac_ctx geoac_ctx;   /* opaque: owns device buffers, stream, events */

/* fill `p` with the reference defaults for an equation set (GeoAc.Parameters{,.Global}.cpp,
 * G2S_{,Global}Spline1D.cpp:22-30); vert_limit is set by geoac_upload_atmo_1d unless given. */
int  geoac_default_params(int eqset, geoac_params* p);

int  geoac_create(geoac_ctx** out, int eqset, int device);
int  geoac_destroy(geoac_ctx* ctx);
/* a second context on the same device sharing src's atmosphere tables (read-only device memory), with src's parameters and options: several
 * independent fans at once (one fan in flight per context).  Valid until src uploads another atmosphere or is destroyed (destroy clones first): a clone
 * whose source has done either fails its next geoac_fan_launch with GEOAC_E_INVALID instead of reading freed or replaced tables. */
int  geoac_clone(geoac_ctx* src, geoac_ctx** out);

/* Launch-plan options (epoch length, kernel variants, overlap): for A/B measurements and the schedule-independence tests - a fan's records
 * never depend on them.  key: one of geoac_option_names() (case-insensitive, an optional "GEOAC_" prefix is ignored), value: its decimal
 * text.  Set before geoac_fan_launch / the atmosphere upload they act on.  The library reads NO environment variable unless
 * GEOAC_DEBUG_ENV=1 is set, in which case GEOAC_<KEY> is applied through this function when a context is created. */
int  geoac_set_option(geoac_ctx* ctx, const char* key, const char* value);
const char* const* geoac_option_names(void);        /* NULL-terminated */

/* use a caller-owned hipStream_t (e.g. torch's current stream) instead of the context's own */
int  geoac_set_stream(geoac_ctx* ctx, void* hip_stream);

/* 1-D atmosphere: n nodes; x = altitude [km] (Cartesian sets) or geocentric radius (Global);
 * T [K], u,v [km/s, already tapered], rho [g/cm^3]; slopes4 = natural-spline slopes of T,u,v,rho
 * (4*n, host-computed: replaces Set_Slopes, G2S_Spline1D.cpp:161-196).  Builds the per-segment
 * coefficient tables and uploads them.  */
int  geoac_upload_atmo_1d(geoac_ctx* ctx, int n, const double* x, const double* T, const double* u,
                          const double* v, const double* rho, const double* slopes4);

/* range-dependent Cartesian atmosphere (GEOAC_EQ_3D_RNGDEP): nx x ny profiles of nz nodes on common z nodes;
 * fields are [nx][ny][nz] row-major (T [K], u,v [km/s, tapered], rho).  Replaces Spline_Multi_G2S + Set_Slopes_Multi
 * (G2S_MultiDimSpline3D.cpp:306-425, 1603-1621): vertical natural splines of f, df/dx, df/dy per node are built here.
 * GEOAC_EQ_GLOBAL_RNGDEP (G2S_GlobalMultiDimSpline3D.cpp:313-431, 1473-1492): x = latitudes, y = longitudes [rad], z = geocentric
 * radius [km]; xy_limits of geoac_params then hold GeoAc_lat_min/lat_max/lon_min/lon_max_limit [rad]. */
int  geoac_upload_atmo_3d(geoac_ctx* ctx, int nx, int ny, int nz, const double* x, const double* y, const double* z,
                          const double* T, const double* u, const double* v, const double* rho);
/* the evaluation table the upload built on the device (geoac_grid_table_size(nx, ny, nz) doubles, layout of geoac_grid_table_eq in
 * geoac_host.h, which is the host restatement of the same construction): set-up check / diagnostics */
int  geoac_grid_table_fetch(geoac_ctx* ctx, double* tab, size_t cap);

int  geoac_set_params(geoac_ctx* ctx, const geoac_params* p);
/* the parameters as they stand (defaults resolved: vert_limit, xy_limits after an atmosphere upload) and the equation set */
int  geoac_get_params(geoac_ctx* ctx, geoac_params* p);
int  geoac_get_eqset(geoac_ctx* ctx, int* eqset);
/* c [km/s], u, v [km/s], rho of the uploaded 1-D atmosphere at abscissa x, evaluated on the host (set-up / reporting only) */
int  geoac_medium_1d(geoac_ctx* ctx, double x, double out[4]);

/* launch angles in degrees, exactly the values of the reference's loop variables theta, phi */
int  geoac_fan_set_angles(geoac_ctx* ctx, int n_rays, const double* theta_deg, const double* phi_deg);

/* integrate the whole fan.  All device work is enqueued on the context's stream; the host drives the
 * epochs (it needs the live-ray count to stop), so the call returns when every ray has finished. */
int  geoac_fan_launch(geoac_ctx* ctx);
int  geoac_fan_sync(geoac_ctx* ctx);
/* condition flags of the last completed launch.  GEOAC_FAN_STEP_LIMIT: some ray exhausted step_limit = ray_limit * int(1 / (10 ds_min))
 * (GeoAc.Solver.cpp:14) before leaving the region or reaching the ground.  The reference treats that as an ordinary leg end
 * (GeoAc_Propagate_RK4 returns step_limit with check = false and the row is written), so the launch succeeds and the leg's record is
 * kept: GEOAC_REC_STEPS = step_limit as the reference returns it (step_limit - 1 steps were taken; the reference's post-pass then
 * reads one row it never wrote - here the sums end at the last integrated row). */
#define GEOAC_FAN_STEP_LIMIT 1
/* GEOAC_FAN_SUB_FALLBACK: a sub-epoch workgroup of the cooperative grid kernels gave up waiting for its predecessor (the in-order workgroup
 * dispatch the hand-off relies on did not hold on this device); the fan was run again without sub-epochs and this context keeps them off.
 * GEOAC_FAN_ABS_FALLBACK: the absorption table left more path segments to the exact fix-up pass than its list holds; the fan was run again
 * with the exact post-pass and this context keeps it.  Both are sticky: reported by every later geoac_fan_status of the context.  Results are
 * the same bits either way (tests); the flags tell an operator that the faster plan is off. */
#define GEOAC_FAN_SUB_FALLBACK 0x100
#define GEOAC_FAN_ABS_FALLBACK 0x200
int  geoac_fan_status(geoac_ctx* ctx, uint64_t* flags);
/* Stratified sets: in a stratified medium SuthBass_Alpha (Atmo_State.Absorption{,.Global}.cpp:12-141) depends on the height coordinate
 * alone, so the post-pass does not evaluate it at every path-segment midpoint: per spline segment a table holds degree-5 interpolants (six
 * coefficients each, Chebyshev nodes) of its three smooth pieces, built on the device from the exact routine and checked against it (the classical term's sqrt(1 + nu^2) - 1 is formed
 * at the midpoint as the reference forms it), and only the path segments the table does not serve are evaluated exactly.  Of the last
 * completed launch: entries of the table (0: table not in use), entries flagged at build time (reassembled alpha off by more than 1e-10
 * relative at a check point), path segments evaluated exactly, largest check-point error of the unflagged entries. */
int  geoac_abs_table_info(geoac_ctx* ctx, int* entries, int* flagged, uint64_t* fixup_segments, double* worst_rel_err);

/* device pointer to the record table [n_rays][bounces+1][GEOAC_REC_STRIDE] f64 (valid after launch,
 * ordered on the context's stream) - what a multi-GPU caller hands to its gather collective */
int  geoac_fan_records_dev(geoac_ctx* ctx, void** dev_ptr, size_t* bytes);
/* asynchronous device-to-device copy of the record table into a caller-owned device buffer (e.g. a torch
 * tensor that is then gathered with RCCL), ordered on the context's stream */
int  geoac_fan_copy_records_dev(geoac_ctx* ctx, void* dst_dev);
/* copy records (and the total step count = sum of GeoAc_Propagate_RK4 return values) to the host */
int  geoac_fan_fetch(geoac_ctx* ctx, double* rec_host, uint64_t* total_steps);
/* WriteRays / WriteCaustics samples: number available, then copy (ordered by ray, leg, m) */
int  geoac_fan_sample_count(geoac_ctx* ctx, int64_t* n);
/* rows the device-side sample list can hold (default 4 Mi, GEOAC_SMP_STRIDE doubles each); a launch that produces more returns
 * GEOAC_E_CAPACITY (geoac_fan_sample_count then tells how many it needed): callers split the fan or raise the capacity */
int  geoac_fan_set_sample_capacity(geoac_ctx* ctx, int64_t rows);
int  geoac_fan_fetch_samples(geoac_ctx* ctx, double* smp_host, int64_t cap);

/* blocking convenience: set_angles + launch + sync + fetch */
int  geoac_fan_run(geoac_ctx* ctx, int n_rays, const double* theta_deg, const double* phi_deg,
                   double* rec_host, uint64_t* total_steps);

/* HIP-event timings of the last completed launch, in ms: [0] whole launch, [1] RK4 kernels, [2] post-pass kernels;
 * and launch statistics: [0] rk4 kernel launches, [1] path bytes written, [2] path bytes read */
int  geoac_last_timing(geoac_ctx* ctx, double ms[3], uint64_t stats[3]);

const char* geoac_strerror(int code);
const char* geoac_last_error(geoac_ctx* ctx);
const char* geoac_version(void);
/* identity of this build: a hash of the sources, headers and flags the library was compiled from (hipcc's output is not bit-reproducible, a file hash is not an
 * identity).  Profiles taken on a build carry it (tools/pmc_derive.py); bench.py marks counter-derived figures stale when the loaded library's differs. */
const char* geoac_build_id(void);
/* 1: an A/B build (make AB=1) that also holds the diagnostic kernels the launch plan never selects (options DUO, GRID_LANES=2); 0: the shipped build */
int         geoac_build_has_ab(void);

#ifdef __cplusplus
}
#endif
#endif /* GEOAC_HIP_H_ */
