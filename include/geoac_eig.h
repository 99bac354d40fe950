/* geoac_eig.h - eigenray searches on top of the ray-fan C ABI (geoac_hip.h): spherical sets and 3-D Cartesian sets.
 *
 * Replaces the callers of the hot path in GeoAc's -eig_search / -eig_direct modes:
 *   GeoAc_EstimateEigenray   Code/GeoAc/GeoAc.Eigenray.Global.cpp:46-136   inclination scans at the great-circle bearing
 *   GeoAc_3DEigenray_LM      Code/GeoAc/GeoAc.Eigenray.Global.cpp:139-319  Newton refinement with the auxiliary (Jacobian) equations
 *   the driver loops         Code/GeoAcGlobal_main.cpp:566-580, Code/GeoAcGlobal.RngDep_main.cpp:604-616
 *   and the Cartesian twins  Code/GeoAc/GeoAc.Eigenray.cpp:30-121, 123-335; Code/GeoAc3D_main.cpp:531-543, Code/GeoAc3D.RngDep_main.cpp
 * The reference traces one ray at a time; here every decision point of every receiver's search asks for the rays it needs and the
 * requests of all receivers are integrated together as ONE fan launch per round (an inclination scan is a single launch of up to
 * (theta_max - theta_min) / d_theta rays instead of that many sequential propagations).  The decisions are then replayed on the host in
 * the reference's order, so the eigenray list and the iteration log are the reference's.  Within a receiver every bounce count is its
 * own scan chain and every refinement its own task (they do not depend on each other), which makes the rounds fewer and fuller.
 * Context requirements: equation set GEOAC_EQ_GLOBAL / _GLOBAL_RNGDEP (receivers = latitude, longitude [deg]) or GEOAC_EQ_3D / _3D_RNGDEP
 * (receivers = x, y [km]; GEOAC_EIG_BEARING then holds the azimuth from the receiver to the source), atmosphere uploaded, parameters set (source position,
 * z_grnd, freq, tweak_abs, limits are taken from geoac_set_params; bounces / calc_amp / mode are managed by the search and restored).
 */
#ifndef GEOAC_EIG_H_
#define GEOAC_EIG_H_

#include "geoac_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    double theta_min, theta_max;   /* inclination range of the search [deg]: 0.5, 45                     */
    int    bnc_min, bnc_max;       /* ground-reflection counts searched: 0, 0                            */
    int    iterations;             /* limit of the refinement iterations: 25                             */
    double azimuth_err_lim;        /* accepted azimuth deviation of the estimate [deg]: 2.0              */
    int    verbose;                /* 1: keep the reference's verbose text per receiver (geoac_eig_log)  */
} geoac_eig_params;

int geoac_eig_default_params(geoac_eig_params* p);

/* one record per identified eigenray */
#define GEOAC_EIG_STRIDE 16
enum {
    GEOAC_EIG_RCVR = 0,      /* receiver index                                                           */
    GEOAC_EIG_INDEX = 1,     /* eigenray number of that receiver (the N of <title>_Eigenray-N.dat)       */
    GEOAC_EIG_BOUNCES = 2,
    GEOAC_EIG_THETA = 3,     /* launch inclination [deg]                                                 */
    GEOAC_EIG_PHI = 4,       /* launch azimuth from north [deg] (the reference prints 90 - lp)           */
    GEOAC_EIG_TTIME = 5,     /* travel time [s]                                                          */
    GEOAC_EIG_CELERITY = 6,  /* great-circle source-receiver distance / travel time [km/s]               */
    GEOAC_EIG_AMP_DB = 7,    /* 20 log10 GeoAc_Amplitude at the arrival                                  */
    GEOAC_EIG_ATTEN_DB = 8,  /* -attenuation [dB]                                                        */
    GEOAC_EIG_INCL = 9,      /* arrival inclination [deg]                                                */
    GEOAC_EIG_BEARING = 10,  /* bearing receiver -> source [deg]                                         */
    GEOAC_EIG_BACKAZ = 11,   /* back azimuth of the arrival [deg]                                        */
    GEOAC_EIG_AZDEV = 12,    /* back azimuth - bearing, wrapped once [deg]                               */
    GEOAC_EIG_NSMP = 13,     /* raypath rows of this eigenray (every 25th step, all legs)                */
    GEOAC_EIG_SMP0 = 14      /* index of its first row in the sample table                               */
};

typedef struct geoac_eig_result geoac_eig_result;   /* opaque, owned by the library until geoac_eig_free */

/* -eig_search for n_rcvr receivers (rcvr = [n_rcvr][2]: latitude, longitude in degrees) around the context's source */
int geoac_eig_search(geoac_ctx* ctx, const geoac_eig_params* p, int n_rcvr, const double* rcvr, geoac_eig_result** out);
/* -eig_direct: refinement only, from a given inclination / azimuth-from-north estimate per receiver, `bounces` reflections */
int geoac_eig_direct(geoac_ctx* ctx, const geoac_eig_params* p, int n_rcvr, const double* rcvr,
                     const double* theta_est, const double* phi_est, int bounces, geoac_eig_result** out);

int64_t     geoac_eig_count(const geoac_eig_result* r);
int         geoac_eig_fetch(const geoac_eig_result* r, double* eig /* count x GEOAC_EIG_STRIDE */);
int64_t     geoac_eig_sample_count(const geoac_eig_result* r);
/* raypath rows of all eigenrays, GEOAC_SMP_STRIDE doubles each (GEOAC_SMP_RAY = index into the eigenray table) */
int         geoac_eig_fetch_samples(const geoac_eig_result* r, double* smp);
/* the reference's verbose text for one receiver ("" unless verbose was set) */
const char* geoac_eig_log(const geoac_eig_result* r, int rcvr);
/* [0] fan launches, [1] rays integrated, [2] RK4 ray-steps, [3] rounds (decision points served) */
int         geoac_eig_stats(const geoac_eig_result* r, uint64_t stats[4]);
/* the same plus [4] critical ray-steps: the sum over the fan launches of the longest ray's step count (a launch lasts as long as its longest
 * ray: what the search costs on a latency-bound device), [5] fan launches that carried the auxiliary (amplitude) equations, [6], [7] reserved */
int         geoac_eig_stats_ex(const geoac_eig_result* r, uint64_t stats[8]);
void        geoac_eig_free(geoac_eig_result* r);

#ifdef __cplusplus
}
#endif
#endif /* GEOAC_EIG_H_ */
