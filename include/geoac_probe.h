/* geoac_probe.h - device-function probes of libgeoac_hip.so (diagnostics and parity tests).
 *
 * The reference exposes its atmosphere as free functions a caller can evaluate anywhere:
 *   c, u, v, rho, c_diff / c_ddiff ... (1-D profiles)            Code/Atmo/G2S_GlobalSpline1D.cpp:332-428, G2S_Spline1D.cpp:321-416
 *   SuthBass_Alpha(z, freq)                                        Code/Atmo/Atmo_State.Absorption{,.Global}.cpp:12-141
 *   Eval_Spline_AllOrder2 and the scalar API of the grid sets      Code/Atmo/G2S_MultiDimSpline3D.cpp:1341-1593, 1633-1743,
 *                                                                  Code/Atmo/G2S_GlobalMultiDimSpline3D.cpp:1224-1461, 1502-1611
 * On the GPU these live inside the RK4 and post-pass kernels.  The probes below run the SAME device functions (seg_eval,
 * suthbass_alpha, grid_eval_all / the cooperative gather grid_eval3_coop, medium3_at), one thread per point, and return what they
 * computed - so the table lookups and the absorption model are checked against the reference's values point by point
 * (tests/test_gpu_probes.py) and not only through whole-fan integrals.  They need the context's parameter block of a completed
 * launch (geoac_fan_launch at least once: reference state of the absorption model, table pointers); host pointers in and out.
 */
#ifndef GEOAC_PROBE_H_
#define GEOAC_PROBE_H_

#include "geoac_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 1-D sets: out9[9 i ..] = c, c', c'', u, u', u'', v, v', v'' and rho[i] at abscissa x[i] (altitude, or geocentric radius for the
 * spherical set); inputs are clamped to the profile like the reference's accessors */
int geoac_probe_atmo_1d(geoac_ctx* ctx, int n, const double* x, double* out9, double* rho);

/* 1-D sets: SuthBass_Alpha(x[i], freq[i]) * tweak_abs * 8.685889 as the post-pass evaluates it (z_grnd and abs_coeff of geoac_set_params) */
int geoac_probe_absorption(geoac_ctx* ctx, int n, const double* x, const double* freq, double* alpha);

/* 1-D sets: the same coefficient from the absorption table the post-pass reads (k_atab_build: per spline segment three degree-5 interpolants - six
 * coefficients each - of the smooth pieces of the routine above at the frequency of geoac_set_params, reassembled with the routine's own
 * roots; accuracy contract: every entry is checked against the exact routine at eight further points to 1e-10 relative, an entry that fails
 * is flagged and the post-pass evaluates the segments that fall into it exactly (k_ppfix); Atmo_State.Absorption{,.Global}.cpp:12-141 tabulated, as alpha depends on the
 * height coordinate alone in a stratified medium).  alpha[i] = -1 where the table does not serve x[i] (flagged segment, beyond the strips
 * at the two ends of the profile): the post-pass evaluates such midpoints with the exact routine. */
int geoac_probe_absorption_table(geoac_ctx* ctx, int n, const double* x, double* alpha);

/* grid sets: point (a0, a1, a2) in table order - (x, y, z) for GEOAC_EQ_3D_RNGDEP, (lat, lon, r) [rad, rad, km] for
 * GEOAC_EQ_GLOBAL_RNGDEP.  out30[30 i + 10 f + q]: field f = T, u, v; q = f, d/da0, d/da1, d/da2, d2/da0^2, d2/da1^2, d2/da2^2,
 * d2/da0 da1, d2/da0 da2, d2/da1 da2 (Eval_Spline_AllOrder2 with the reference's quirks Q11 / Q12).  api7[7 i ..] = c, rho, u, v and
 * d/da2 of c, u, v through the scalar evaluators.  coop != 0: through the wave-cooperative gather the dense grid fans use. */
int geoac_probe_grid(geoac_ctx* ctx, int n, const double* a0, const double* a1, const double* a2, int coop, double* out30, double* api7);

#ifdef __cplusplus
}
#endif
#endif /* GEOAC_PROBE_H_ */
