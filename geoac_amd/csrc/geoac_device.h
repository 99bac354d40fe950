// geoac_device.h - shared host/device definitions of the MI355X ray-fan integrator.
//
// Data layout in HBM (all FP64 unless noted):
//   seg table   [nseg][SEGW]            per spline segment: x0, x1, then the cubic of T, u, v in t = x - x0 as
//                                       (c0, c1, 2 c2, 6 c3).  Staged into LDS by the RK4 kernel (112 B/segment; ToyAtmo
//                                       1399 segments = 153 KiB of the CU's 160 KiB).
//   rho table   [nseg][4]               cubic coefficients of density (needed only at arrivals / in the post-pass)
//   state       [NSTATE][n_pad]         per-ray persistent state, SoA (ray index fastest => coalesced)
//   path chunk  [S_rows][PATHW][n_pad]  the state rows the post-pass needs (Global: r,lat,lon,nu_r,nu_t,nu_p),
//                                       one row per accepted RK4 step, ray index fastest => every wave store is a
//                                       contiguous 512 B line.  This is the reference's `double** solution`
//                                       (Interface.cpp:53-58) re-laid-out for 64-wide coalescing and cut into epochs.
//   contrib     [S_rows][2][n_pad]      per-segment travel-time and attenuation increments (post-pass kernel)
//   records     [n_rays][legs][32]      one arrival record per (ray, leg), layout in include/geoac_hip.h
#ifndef GEOAC_DEVICE_H_
#define GEOAC_DEVICE_H_

#include <stdint.h>

#define GEOAC_SEGW      14      // doubles per segment in the T/u/v table
#define GEOAC_MAXE      18
#define GEOAC_MAXLEGS   64      // legs per ray supported by the per-epoch leg-end event list
#define GEOAC_ATABW     20      // doubles per entry of the absorption table: 2 / h (negative: flagged), 3 x 6 coefficients, worst check-point error
#define GEOAC_LAT_OFF   1024    // lat_trig: entry of latitude 0
#define GEOAC_LAT_N     2049    // ... entries (|lat| <= 8 rad; the index is clamped)
#define GEOAC_PP_ROWS   16      // path segments per thread of k_postpass_tab (consecutive rows of one ray: each row is read once)
#define GEOAC_CNT_PPFLAG 28     // counters[+0]: entries of the fix-up list of the current k_postpass_tab launch; counters[+1]: path segments of the fan the absorption table did not serve (evaluated exactly by k_ppfix)

// per-ray state slots (SoA rows of the state buffer)
enum {
    ST_Y0     = 0,      // y[0..17]
    ST_K      = 18,     // steps taken in the current leg (as double, exact)
    ST_LEG    = 19,     // current leg index
    ST_DONE   = 20,     // 1.0 when the ray is finished
    ST_HMAX   = 21,     // running turning height
    ST_C0     = 22,     // sound speed at the source
    ST_NU0    = 23,     // 1/MachScalar (Global) ; c_eff_0 (2D)
    ST_AUX0   = 24,     // eqset-specific constants (3D: nu0_xy[2], mu0_xy[2][2] => 6 values; 2D: cos/sin phi)
    ST_SEG    = 30,     // segment hint (as double)
    ST_TT     = 31,     // post-pass running travel time (owned by the accumulate kernel)
    ST_AT     = 32,     // post-pass running attenuation
    ST_PLEG   = 33,     // post-pass current leg
    ST_LTT    = 34,     // post-pass per-leg partial sums (arrivals-only form of Q7)
    ST_LAT    = 35,
    ST_YM2    = 36,     // row k-2 of the current leg (Cartesian sets: quadratic ground intercept), up to 18 values (3D.RngDep)
    ST_DPREV  = 54,     // Jacobian of the previous row (WriteCaustics)
    ST_NSTATE = 55
};

struct GeoacDevParams {
    // problem
    int     eqset, calc_amp, mode, bounces;
    int     n_rays, n_pad;          // n_pad = n_rays rounded up to 64
    int     E, pathw;
    int     nseg;                   // spline segments = nodes - 1
    int     s_rows;                 // path rows per epoch chunk
    int     table_in_lds;
    int     live_slot;              // index into counters[] of the live-ray count this launch adds to (1, or 6 for the second launch of a hybrid fan)
    int     slot_lo, slot_hi;       // k_rk4 integrates the ray slots [slot_lo, slot_hi) (a fan may be split over two concurrent launches)
    int     lanes_per_ray;          // 2: Global + CalcAmp without sample capture runs the two-lanes-per-ray kernel
    int     duo;                    // 1: the wave-specialised kernel k_rk4_duo (geoac_duo.h): one wave integrates 64 rays, a second one their
                                    // launch-angle derivative systems from per-stage messages in LDS (lanes_per_ray = 1: the one-lane state layout)
    int     trio;                   // 1: slots the plan gives two lanes per ray run the wave-specialised kernel k_rk4_trio instead (geoac_trio.h: the ray on one wave, one
                                    // launch-angle system on each of two more; the same state layout)
    int     spread;                 // grid sets: only every spread-th lane of a wave carries a ray (power of two, 1..64): a small fan is
                                    // spread over more waves so that each divergent table gather touches fewer cache lines per instruction
    int     quad_cache;             // grid sets, four lanes per ray, at most 256 waves: per-lane record cache and z nodes in LDS (grid_cache_fill)
    int     coop;                   // grid sets, one lane per ray: wave-cooperative record gather through LDS (grid_eval3_coop)
    int     seg_safe;               // 1: every spline segment is longer than the largest RK4 step => the +-1 segment move is exact
    int     pp_blocks;              // grid size of the persistent post-pass kernel
    int     rays_form;              // post-pass sums in the WriteRays form (segments 0..k-2, cumulative): Q7
    long long step_limit;           // GeoAc.Solver.cpp:14
    double  x_min, x_max;           // clamp range of the spline abscissa
    double  ds_min, ds_max;
    double  ground;                 // Global: r_earth + z_grnd ; Cartesian: z_grnd
    double  r_earth, z_grnd;
    double  vert_limit, range_limit, range_thresh;   // range_thresh: sin^2(range_limit/(2 r_earth)) (Global)
    double  range_sq[2];            // 3-D stratified set: range_limit^2 (1 -/+ 1e-12): the horizontal range is compared squared outside this band (Eq3D::checks)
    double  range_skip;             // Global: while (|lat - lat_src| + |lon - lon_src|) / 2 stays below this, the range test cannot fire (EqGlobal::checks)
    double  src[3];                 // as in geoac_params
    double  freq, tweak_abs;
    double  T_o, P_o;               // SuthBass reference temperature / pressure (ground), host-evaluated from the spline
    double  cbrt_To;                // cbrt(T_o)
    double  c000;                   // c(0,0,0) used by the 3-D travel-time integral (Q5)
    double  src_trig[2];            // Global: sin, cos of the source latitude
    double  sb_const[5];            // 10^-0.67887, 10^-0.10744, 10^-3.3979, 5/sqrt(21), sqrt(3/7) (host libm, as the reference computes them)
    // buffers
    const double* seg;              // [nseg][SEGW]
    const double* rho;              // [nseg][4]
    // stratified sets: absorption table (k_atab_build): Sutherland-Bass alpha is a function of the height coordinate alone there, so it is
    // tabulated per spline segment instead of being evaluated at every path-segment midpoint (k_postpass_tab)
    int*    ppfix;                  // fix-up list of k_postpass_tab: (column, chunk row) of the segments the table did not serve, ppfix_cap pairs (k_ppfix evaluates them exactly)
    int     ppfix_cap;
    int     pp_lds_pad;             // bytes of LDS a k_postpass_tab workgroup asks for without using them: keeps it off CUs that hold an RK4 workgroup (> 7 KiB) / to one workgroup per CU (> 80 KiB)
    int     accum_batch;            // k_accum: fetch the contributions of eight rows together (late epochs of arrivals-only fans: the sums are the fan's uncovered tail)
    int     pp_lds_table;           // k_postpass_tab<.., TBL> (spherical set, one-trip form): the table entry in hand in LDS instead of 38 registers, no row prefetch: 127 registers, four waves per SIMD
    int     pp_onetrip;             // k_postpass_tab: on a change of spline segment fetch the neighbour's record and table entry in one trip (fans that fill the chip); 0: the walk
    const double* atab;             // [nseg + 2][GEOAC_ATABW]: 2/h (negative: flagged), 3 x six coefficients in s = 2 t / h - 1 (atab_eval); entries nseg / nseg + 1: the
                                    // strips of width atab_D below the first / above the last node (medium clamped there, height not)
    const double* lat_trig;         // spherical set: [GEOAC_LAT_N][2] sin, cos of the multiples of 2^-7 rad (index i: (i - GEOAC_LAT_OFF) / 128), behind the table in the same buffer
                                    // (k_atab_build): the reference points pp_geom rotates the midpoint latitudes' sin / cos from - read, not recomputed, in k_postpass_tab
    int           atab_on;          // 1: k_postpass_tab + fix-up of the flagged segments; 0: exact evaluation at every midpoint (k_postpass)
    double        atab_D;           // width of the two strips
    double        atab_lo;          // x_min - atab_D: left end of the lower strip
    double        seg_per_x;        // nseg / (x_max - x_min): first guess of the segment index without a division
    // range-dependent sets: grid of profiles
    int           gnx, gny;         // horizontal node counts
    double        g_lo[2], g_hi[2]; // first / last node of gx and gy (clamp range of the horizontal coordinates; kernel arguments: no loads)
    const double* gx;               // [gnx] node x
    const double* gy;               // [gny] node y
    const double* gz;               // [nseg+1] node z (x_min / x_max hold the z range)
    const double* gtab;             // grid sets: T, u, v [3][nseg][gnx*gny][40 | 32 packed, Cartesian] then rho [nseg][gnx*gny][16] (geoac_rngdep.h)
    double        xy_lim[4];        // x_min, x_max, y_min, y_max break limits (GeoAc.Parameters.RngDep.cpp:24-28)
    double*       dev_consts;       // [0] T_o, [1] P_o, [2] cbrt(T_o) of SuthBass evaluated on the device (RngDep: medium at (0, 0, z_grnd))
    const int*    colmap;           // live-ray compaction: column of this epoch's chunk -> ray slot (NULL: identity, column = slot)
    const int*    n_cols;           // device: number of valid columns of colmap
    int           n_cols_bound;     // host-side upper bound of the columns in use this epoch (launch sizes of the post-pass / sum kernels); n_pad without compaction
    int           sub;              // cooperative grid kernels: the epoch is cut into `sub` sub-epochs, each a workgroup of its own (k_rk4); 1 = off
    int           sub_test_stall;   // tests (SUB_TEST_STALL): sub-epoch 0 never publishes its flag and the waiters give up after 20 ms - forces the hand-off time-out path
    int           sub_w;            // workgroups per sub-epoch (multiple of 8: a wave's sub-epochs are dispatched on one XCD, in order)
    int*          sub_flags;        // [sub_w] sub-epochs completed per wave (zeroed before the launch)
    const int*    perm;             // slot -> ray index of the caller's order (records and samples are written in the caller's order); NULL = identity
    const double* theta_deg;        // [n_rays]
    const double* phi_deg;
    double*       state;            // [ST_NSTATE][n_pad]
    double*       path;             // [s_rows][pathw][n_pad]
    double*       contrib;          // [s_rows][2][n_pad]
    int*          nrows;            // [n_pad] rows written by each ray in the current chunk
    int*          legend;           // [GEOAC_MAXLEGS][n_pad] chunk-row index of leg-end rows in this chunk
    int*          nlegend;          // [n_pad]
    // WriteRays / WriteCaustics events of the current chunk (k_rk4 -> k_accum) and the sample output list
    int           smp_stride, ev_cap;
    int*          ev_row;           // [ev_cap][n_pad] chunk-row index of the event
    int*          ev_m;             // [ev_cap][n_pad] row index m within the leg | kind << 30
    double*       ev_amp;           // [ev_cap][n_pad] GeoAc_Amplitude at the row (raypath rows)
    int*          nev;              // [n_pad]
    double*       smp_out;          // [smp_cap][GEOAC_SMP_STRIDE]
    long long     smp_cap;
    double*       rec;              // [n_rays][bounces+1][32]
    unsigned long long* counters;   // [0] total steps, [1] active rays after this epoch, [2] error flags, [3] samples emitted
};

#endif
