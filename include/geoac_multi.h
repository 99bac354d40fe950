/* geoac_multi.h - one launch-angle fan over several GPUs of a node, from one process (libgeoac_hip.so).
 *
 * The reference integrates the fan in a serial double loop over azimuth and inclination
 * (Code/GeoAcGlobal_main.cpp:241-242, Code/GeoAc3D.RngDep_main.cpp:244-245 and twins); rays are independent, so the loop shards by
 * azimuth with no exchange during the integration.  A pool owns one geoac_ctx per listed device (each holds the whole atmosphere:
 * 157 KB for a profile, 38 MB for a 5x5x1400 grid) and one host thread per context.  geoac_pool_fan_run cuts the fan into azimuth
 * groups (consecutive rays of equal azimuth, the reference's outer loop), hands them out from a shared queue - the devices balance
 * themselves however the cost varies with azimuth - and every group's arrival records are copied device -> host straight into the
 * rows of the caller's table: the gather is those copies, no collective is needed inside one process.  (The one-process-per-GPU form,
 * torch.distributed + RCCL all_gather of the record tables, is geoac_amd/sharding.py / bench.py.)
 * Results do not depend on the device list, its order or the group size: a ray's records are those of the same ray integrated alone.
 */
#ifndef GEOAC_MULTI_H_
#define GEOAC_MULTI_H_

#include "geoac_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct geoac_pool geoac_pool;

/* one context per entry of devices[] (an index may repeat: two contexts on one GPU) */
int  geoac_pool_create(geoac_pool** out, int eqset, int n_dev, const int* devices);
int  geoac_pool_destroy(geoac_pool* pool);
int  geoac_pool_size(const geoac_pool* pool);
/* the i-th context, e.g. for geoac_get_params or the probes; set-up calls should go through the pool so that every device holds the same state */
geoac_ctx* geoac_pool_ctx(geoac_pool* pool, int i);

/* the set-up calls of geoac_hip.h, applied to every context */
int  geoac_pool_upload_atmo_1d(geoac_pool* pool, int n, const double* x, const double* T, const double* u, const double* v,
                               const double* rho, const double* slopes4);
int  geoac_pool_upload_atmo_3d(geoac_pool* pool, int nx, int ny, int nz, const double* x, const double* y, const double* z,
                               const double* T, const double* u, const double* v, const double* rho);
int  geoac_pool_set_params(geoac_pool* pool, const geoac_params* p);

/* geoac_fan_run over the pool's devices (arrivals-only modes: sample capture needs the single-context calls).  rays_per_group <= 0:
 * automatic (whole azimuths, about four groups per device, at least 16 384 rays per group while the fan allows).  rec_host:
 * [n_rays][bounces+1][GEOAC_REC_STRIDE] in the caller's ray order. */
int  geoac_pool_fan_run(geoac_pool* pool, int n_rays, const double* theta_deg, const double* phi_deg, int rays_per_group,
                        double* rec_host, uint64_t* total_steps);
/* [i] = rays / ray-steps / groups integrated by context i in the last geoac_pool_fan_run (load balance) */
int  geoac_pool_last_shares(const geoac_pool* pool, uint64_t* rays, uint64_t* steps, uint64_t* groups);
/* geoac_fan_status flags of the last geoac_pool_fan_run, OR-ed over every group of every device (GEOAC_FAN_STEP_LIMIT) */
int  geoac_pool_fan_status(const geoac_pool* pool, uint64_t* flags);
const char* geoac_pool_last_error(const geoac_pool* pool);

#ifdef __cplusplus
}
#endif
#endif /* GEOAC_MULTI_H_ */
