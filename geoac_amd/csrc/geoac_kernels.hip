// geoac_kernels.hip - hand-written gfx950 kernels of the ray-fan integrator.
//
// Four kernels per epoch (an epoch advances every live ray by up to s_rows-2 RK4 steps):
//   k_init      once per fan: launch-angle -> initial conditions            (GeoAc_SetInitialConditions)
//   k_rk4       one ray per lane, register-resident FP64 state, spline segment table in LDS; per step:
//               4 x {sources + RHS} (GeoAc_UpdateSources/GeoAc_EvalSrcEq fused and specialised for the
//               stratified atmosphere), ds rule, break/ground checks, on-device bounce handling and arrival
//               records; writes one coalesced state row per step into the path chunk   (GeoAc_Propagate_RK4 + the
//               launch-angle / bounce loops of the mains)
//   k_postpass  one thread per path SEGMENT: midpoint travel-time and Sutherland-Bass absorption increments
//               (GeoAc_TravelTime / GeoAc_SB_Atten / SuthBass_Alpha) - the transcendental-heavy part, taken off the
//               serial RK4 recurrence and run at full chip occupancy
//   k_accum     one thread per ray: in-order summation of the increments (the reference's summation order),
//               leg bookkeeping (Q7 forms), fills the travel-time / attenuation fields of the records
//
// No MFMA: there is no dense contraction anywhere on this path.
#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>
#include "geoac_device.h"
#include "../../include/geoac_hip.h"

#define DEVINL __device__ __forceinline__
#ifndef GEOAC_CACHE_REG_STATE
#define GEOAC_CACHE_REG_STATE 0       // record-cache kernels (four lanes per ray, small fans): the step rows in registers (1) or in LDS (0; measured: the 64-receiver ring 4.4 s against 4.7 s with 1)
#endif
#ifndef GEOAC_OCT_LDS_STATE
#define GEOAC_OCT_LDS_STATE 0         // eight-lane grid kernels: the step's rows in LDS (1) or in registers (0: since the arrival evaluation left k_rk4 they fit)
#endif
#define GEOAC_ROT_MAX 1.0e-3             // largest angle rot_small is asked to rotate by (EqGlobal::checks)

// Floating-point contraction: OFF for the stratified sets' code (from here to the include of geoac_rngdep.h, and again from geoac_duo.h
// on) - the only fused multiply-adds are the ones written out as __builtin_fma.  hipcc's default (fast) lets the optimiser fuse a
// product into a following sum wherever it sees one, and what it sees depends on the kernel a function is inlined into: the same source
// then gives different bits in k_rk4<EqGlobal>, k_rk4<EqGlobalPair> and k_rk4_duo, and records must not depend on the launch plan.
#pragma clang fp contract(off)

#ifndef GEOAC_RCPC
#define GEOAC_RCPC 1                     // 1/r and 1/cos(lat) of stages 1-3 by a Newton step from stage 0's values (global_base); 0: a fresh seed per stage (A/B)
#endif
#ifndef GEOAC_AB
#define GEOAC_AB 0                      // 1: A/B build (`make AB=1`) - also holds the diagnostic kernels the launch plan never selects
#endif

static constexpr double kPi   = 3.141592653589793238462643;   // GeoAc.Parameters.cpp:27
static constexpr double kGam  = 1.4;
static constexpr double kRgas = 287.05;
static constexpr double kGamR = 0.00040187;                    // G2S_GlobalSpline1D.cpp:343

// ------------------------------------------------------------------------------------------------
// small FP64 helpers
// ------------------------------------------------------------------------------------------------
// v_rcp_f64 / v_rsq_f64 are ~24-bit seeds on gfx950 (measured 4.6e-8 / 5.2e-8 max rel. error, tools/ubench_acc.hip);
// two Newton steps bring both to <= 1 ulp (measured 0 / 2.6e-16) at 24 / 35 ns per wave-instruction group instead of
// the 33 / 49+33 ns of the IEEE division / sqrt+division expansions (tools/ubench_fp64.hip, one wave per SIMD).
DEVINL double frcp(double x){
    // v_rcp_f64 is a 24-bit seed on gfx950: one third-order step (r (1 + e + e^2), e = 1 - x r) leaves e^3 ~ 2^-72
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    double t = __builtin_fma(e, e, e);
    return __builtin_fma(r, t, r);
}
DEVINL double frsq(double x){
    // y (1 - e)^(-1/2) = y (1 + e/2 + 3 e^2 / 8 + ...), e = 1 - x y^2: one third-order step from the 24-bit seed
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double e = __builtin_fma(-g, y, 1.0);
    double p = __builtin_fma(e, 0.375, 0.5);
    return __builtin_fma(y * e, p, y);
}
// exp / exp10 for the absorption integrand (k_postpass): arguments of moderate size and never NaN / inf there, so none of the library
// routine's special-case work.  k = round(x log2 e), r = x - k ln2 (two-constant Cody-Waite, exact product through FMA), Taylor
// polynomial of degree 13 on |r| <= ln2 / 2 (truncation 2e-17), result scaled by 2^k: < 1 ulp + the rounding of r.
DEVINL double exp_poly(double r){
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_fma(p, r, 1.0);
}
DEVINL double fexp(double x){
    const double k = __builtin_rint(x * 1.44269504088896338700e+00);
    double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
    return __builtin_ldexp(exp_poly(r), (int)k);
}
DEVINL double fexp10(double y){
    const double k = __builtin_rint(y * 3.32192809488736218171e+00);
    double r = __builtin_fma(-k, 3.01029995663611771306e-01, y);       // log10(2), high and low parts (fdlibm's log10_2hi / lo)
    r = __builtin_fma(-k, 3.69423907715893078616e-13, r);
    return __builtin_ldexp(exp_poly(r * 2.30258509299404568402e+00), (int)k);
}
// GeoAc_Set_ds (Global.cpp:210-217 and twins): 0.05 - 0.049 exp(-h / 0.75), clamped to [ds_min, ds_max]; h = height above the ground.
// -h / 0.75 as a product with the double next to -4/3 (the quotient differs by at most one unit in the last place; the IEEE division
// expansion is twelve instructions and a transcendental seed on the serial chain of every step)
// Above 28.5 km the exponential is below 3.2e-17 and 0.049 e below half the spacing of the doubles at 0.05 (3.5e-18): the difference IS 0.05, and
// a wave whose rays are all up there skips the exponential (the same bits either way).
// (Measured and dropped, round 3: exp(-h / 0.75) as a table value at the 1/64 km marks times a degree-6 polynomial - 25 instructions less, 2.5 % MORE
// time, with the table value loaded at the top of the step or a step ahead: the constants the full routine moves into place between its dependent
// multiply-adds fill issue slots that would stay empty anyway, the load and its wait do not.)
DEVINL double set_ds(double h, double ds_min, double ds_max){
    // (round 4: no skip of the exponential above 28.5 km any more - there 0.049 e is below half the spacing of the doubles at 0.05 and the difference IS 0.05, the
    //  same bits with or without it - : the skip was a branch, and a branch ends the scheduling region; without it the exponential's dependent chain sits in the block
    //  of stage 0 and interleaves with that stage's own chains.  Metric pass 120.6 -> 117.6 ms together with the items in seg_locate / open_step, A/B in turn against
    //  the round-3 build; config 3, whose waves are mostly up there, is unchanged: 355-357 ms with and without, profiles/r04_b_cfg3_ab.txt)
    const double ds = 0.05 - 0.049 * fexp(h * (-1.0 / 0.75));
    return __builtin_fmax(__builtin_fmin(ds, ds_max), ds_min);
}

// the same for |d| <= 4e-3 rad: two more terms (d^7 / 5040 < 2e-20)
DEVINL void rot_fifth(double sa, double ca, double d, double& s, double& c){
    const double d2 = d * d;
    const double sd = d * __builtin_fma(d2, __builtin_fma(d2, 1.0 / 120.0, -1.0 / 6.0), 1.0);
    const double cd = __builtin_fma(d2, __builtin_fma(d2, __builtin_fma(d2, -1.0 / 720.0, 1.0 / 24.0), -0.5), 1.0);
    s = __builtin_fma(sa, cd, ca * sd);
    c = __builtin_fma(ca, cd, -sa * sd);
}
// rotate (sin a, cos a) by a small angle d.  |d| <= ds_max / r_earth < 1e-4 for every caller (one RK4 stage or step of at most
// 0.5 km at r >= 6370 km), so sin d = d (1 - d^2/6), cos d = 1 - d^2/2 (1 - d^2/12) are exact to < 1e-17 relative
DEVINL void rot_small(double sa, double ca, double d, double& s, double& c){
    double d2 = d * d;
    double sd = d * __builtin_fma(d2, -1.0 / 6.0, 1.0);
    double cd = __builtin_fma(d2 * (-0.5), __builtin_fma(d2, -1.0 / 12.0, 1.0), 1.0);
    s = __builtin_fma(sa, cd, ca * sd);
    c = __builtin_fma(ca, cd, -sa * sd);
}

// sin & cos together.  Cody-Waite reduction by pi/2 (3 constants, exact products through FMA) and the
// classic minimax kernels on [-pi/4, pi/4]; < 1 ulp for |x| < 1e5, which covers latitudes / longitudes / launch angles.
DEVINL void fsincos(double x, double& s, double& c){
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_1  = 1.57079632673412561417e+00;
    const double pio2_1t = 6.07710050650619224932e-11;
    const double pio2_2t = 2.02226624879595063154e-21;
    double fn = __builtin_rint(x * two_over_pi);
    double r  = __builtin_fma(-fn, pio2_1, x);
    r = __builtin_fma(-fn, pio2_1t, r);
    r = __builtin_fma(-fn, pio2_2t, r);
    int n = (int)fn;
    double z = r * r;
    // sin kernel
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    double sn = __builtin_fma(r * z, ps, r);
    // cos kernel
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    double cs = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    double so = (n & 1) ? cs : sn;
    double co = (n & 1) ? sn : cs;
    s = (n & 2) ? -so : so;
    c = ((n + 1) & 2) ? -co : co;
}

// ------------------------------------------------------------------------------------------------
// spline segment table access.  Record (SEGW doubles): x0, x1, T{c0..c3}, u{c0..c3}, v{c0..c3};
// value = c0 + t (c1 + t (c2 + t c3)), t = x - x0.  Same cubic as Eval_Spline_f (G2S_GlobalSpline1D.cpp:259-295)
// expanded in powers of t on the host (geoac_api.cpp: build_tables).
// ------------------------------------------------------------------------------------------------
struct Atm9 { double T, dT, ddT, u, du, ddu, v, dv, ddv; };

template <typename TabPtr>
DEVINL int seg_find(TabPtr tab, int nseg, double x, int k){
    // hinted walk: rays move <= 50 m per stage while nodes are ~100 m apart, so this loop almost never iterates.
    // Stateless w.r.t. exact node ties (the spline is C2: either neighbour gives the same value to rounding, Q13).
    k = k < 0 ? 0 : (k > nseg - 1 ? nseg - 1 : k);
    while(k > 0 && x < tab[k * GEOAC_SEGW]) k--;
    while(k < nseg - 1 && x > tab[k * GEOAC_SEGW + 1]) k++;
    return k;
}

// cubic in the derivative-friendly form (c0, c1, d2 = 2 c2, e3 = 6 c3): f'' = d2 + e3 t, f' = c1 + t/2 (d2 + f''),
// f = c0 + t (c1 + t/6 (2 d2 + f'')) - 6 FMA-class operations per function for value + both derivatives
DEVINL void cubic3(double c0, double c1, double d2, double e3, double t, double th, double t6, double& f, double& f1, double& f2){
    f2 = __builtin_fma(t, e3, d2);
    const double s2 = d2 + f2;
    f1 = __builtin_fma(th, s2, c1);
    f  = __builtin_fma(t, __builtin_fma(t6, d2 + s2, c1), c0);
}

template <typename TabPtr>
DEVINL void seg_eval_at(TabPtr p, double x, Atm9& a){
    const double t = x - p[0];
    const double th = 0.5 * t, t6 = t * (1.0 / 6.0);
    cubic3(p[2],  p[3],  p[4],  p[5],  t, th, t6, a.T, a.dT, a.ddT);
    cubic3(p[6],  p[7],  p[8],  p[9],  t, th, t6, a.u, a.du, a.ddu);
    cubic3(p[10], p[11], p[12], p[13], t, th, t6, a.v, a.dv, a.ddv);
}

template <bool D2, typename TabPtr>
DEVINL void seg_eval(TabPtr tab, int k, double x, Atm9& a){ seg_eval_at(tab + k * GEOAC_SEGW, x, a); }

// RK4-stage lookup: `off` = element offset (segment index * SEGW) of the segment used by the previous stage.  Rays move
// <= 50 m per stage while nodes are ~100 m apart, so one branch-free +-1 move (two dependent LDS round trips) finds the
// segment; the walk is the never-taken fallback for pathological (very short) segments.
// The whole record of the segment used by the previous stage is fetched at once: the ray is still inside it in all but one
// stage of some tens (<= 50 m per stage against ~100 m between nodes, far less for shallow rays), so the usual stage costs ONE
// LDS round trip instead of two dependent ones (bounds, then coefficients).  seg_fetch issues the loads; the caller puts the
// table-independent part of its right-hand side between seg_fetch and seg_resolve, which hides that round trip as well.
// W = doubles per record of the table `tab`: GEOAC_SEGW (x0, x1, 12 coefficients: the table in global memory and its plain LDS copy) or
// 13 (x0, 12 coefficients; x1 is the next record's x0: the packed LDS copy of k_rk4_duo, which needs the room for its message slots).
// Either way 14 consecutive doubles, returned in the order x0, x1, coefficients.
template <int W = GEOAC_SEGW, typename TabPtr>
DEVINL void seg_fetch(TabPtr tab, int off, double* r){
    const auto* p = tab + off;
    if(W == GEOAC_SEGW){
        #pragma unroll
        for(int c = 0; c < GEOAC_SEGW; c++) r[c] = p[c];
    } else {
        r[0] = p[0]; r[1] = p[13];
        #pragma unroll
        for(int c = 0; c < 12; c++) r[2 + c] = p[1 + c];
    }
}
template <int W = GEOAC_SEGW, typename TabPtr>
DEVINL void seg_resolve(TabPtr tab, const GeoacDevParams& P, double x, int& off, double* r, Atm9& a){
    const int last = (P.nseg - 1) * W;
    const bool up = (x > r[1]) & (off < last), down = (x < r[0]) & (off > 0);
    if(up | down){                                               // rare, divergent: step to the neighbour and fetch again
        off += (up ? W : 0) - (down ? W : 0);
        if(!P.seg_safe){                                         // wave-uniform: only for profiles with nodes closer than one step
            const auto* p = tab + off;
            const double x0 = p[0], x1 = p[W == GEOAC_SEGW ? 1 : 13];
            if(((x < x0) & (off > 0)) | ((x > x1) & (off < last))){
                int k = off / W;
                k = k < 0 ? 0 : (k > P.nseg - 1 ? P.nseg - 1 : k);
                while(k > 0 && x < tab[k * W]) k--;
                while(k < P.nseg - 1 && x > tab[k * W + (W == GEOAC_SEGW ? 1 : 13)]) k++;
                off = k * W;
            }
        }
        seg_fetch<W>(tab, off, r);
    }
    seg_eval_at(r, x, a);
}
template <typename TabPtr>
DEVINL void seg_step_eval(TabPtr tab, const GeoacDevParams& P, double x, int& off, Atm9& a){
    double r[GEOAC_SEGW];
    seg_fetch(tab, off, r);
    seg_resolve(tab, P, x, off, r, a);
}
// RK4-stage lookup with the record kept in registers (r[14], filled by seg_fetch when the ray's state is loaded): a ray stays inside a segment
// for many stages (<= 50 m per stage against ~100 m between nodes; tens of steps for the shallow rays that set a fan's run time), so the usual
// stage touches no memory at all - the bounds are in hand - and the divergent re-fetch happens once per segment crossing
// the two halves of it: make r the record of the segment x lies in (the rare, divergent part) / evaluate it
template <int W = GEOAC_SEGW, typename TabPtr>
DEVINL void seg_locate(TabPtr tab, const GeoacDevParams& P, double x, int& off, double* r){
    const int last = (P.nseg - 1) * W;
    // x is clamped to [x_min, x_max] = [x0 of the first record, x1 of the last] by every caller: x > r[1] cannot hold in the last segment nor
    // x < r[0] in the first - no index guards (four instructions per stage less)
    const bool up = (x > r[1]), down = (x < r[0]);
    (void)last;
    if(__builtin_expect(up | down, 0)){
        off += (up ? W : 0) - (down ? W : 0);
        if(!P.seg_safe){
            const auto* p = tab + off;
            const double x0 = p[0], x1 = p[W == GEOAC_SEGW ? 1 : 13];
            if(((x < x0) & (off > 0)) | ((x > x1) & (off < last))){
                int k = off / W;
                k = k < 0 ? 0 : (k > P.nseg - 1 ? P.nseg - 1 : k);
                while(k > 0 && x < tab[k * W]) k--;
                while(k < P.nseg - 1 && x > tab[k * W + (W == GEOAC_SEGW ? 1 : 13)]) k++;
                off = k * W;
            }
        }
        seg_fetch<W>(tab, off, r);
        // (measured, round 3: waiting for the whole record here, so that the common path carries no counted waits, costs 7 % - the uses' own waits
        //  let the first cubic start while the last pieces are still on their way)
    }
}
template <int W = GEOAC_SEGW, typename TabPtr>
DEVINL void seg_cached_eval(TabPtr tab, const GeoacDevParams& P, double x, int& off, double* r, Atm9& a){
    seg_locate<W>(tab, P, x, off, r);
    seg_eval_at(r, x, a);
}

DEVINL double clampd(double x, double lo, double hi){ double e = (hi < x) ? hi : x; return (e < lo) ? lo : e; }
// the same through v_min_f64 / v_max_f64 (two instructions instead of six; no NaN reaches it): the stage abscissa of the 1-D kernels.  The grid
// kernels keep the select form: min / max canonicalise operands that come straight from memory, and those kernels have no register to spare
DEVINL double clampq(double x, double lo, double hi){ return __builtin_fmax(__builtin_fmin(x, hi), lo); }

// first guess of the segment index for an arbitrary x (uniform-grid guess + walk)
template <typename TabPtr>
DEVINL int seg_guess(TabPtr tab, const GeoacDevParams& P, double x){
    double span = P.x_max - P.x_min;
    int k = (int)((x - P.x_min) / span * (double)P.nseg);
    return seg_find(tab, P.nseg, x, k);
}

// density at abscissa x (global-memory table; rare / post-pass only)
DEVINL double rho_eval(const GeoacDevParams& P, int k, double x){
    const double* s = P.seg + (size_t)k * GEOAC_SEGW;
    const double* q = P.rho + (size_t)k * 4;
    double t = x - s[0];
    return __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, q[3], q[2]), q[1]), q[0]);
}

// ------------------------------------------------------------------------------------------------
// Global equation set: fused GeoAc_UpdateSources + GeoAc_EvalSrcEq (EquationSets.Global.cpp:222-442),
// specialised for the stratified atmosphere (w = 0, every d/dlat, d/dlon of the medium = 0).
// y: r, lat, lon, nu_r, nu_t, nu_p | R_lt(3), mu_lt(3) | R_lp(3), mu_lp(3)
// sth0/cth0: sin/cos of the latitude at the start of the step, dlat: latitude increment of this stage (small-angle rotation).
// ------------------------------------------------------------------------------------------------
// What the launch-angle derivative systems need from the base ray's right-hand side at one RK4 stage: everything else they use follows
// from these by a few multiplications (GlobalDerived).  17 doubles: the per-stage message of the wave-specialised kernel (k_rk4_duo),
// where one wave integrates the base rays and another the derivative systems.
struct GlobalStage { double n0, n1, n2, inm, cn, icg, dc, du, dv, v, cg2, ir, ico, sth, cth, H0, K2; };
#define GEOAC_GSTAGE_W 17

// Base ray: sources + right-hand side of r, lat, lon, nu_r, nu_t, nu_p (Global.cpp:222-272, 369-390).  S: the stage values the derivative
// systems read (K2 only with AMP).
struct NoHook { DEVINL void operator()() const {} };
// HOOK: called once the segment record has arrived (k_rk4_duo reads the consumed-message counter there, so that the answer is back
// when the stage's message is published); ROT0: stage 0 of a step - the stage latitude IS the step's, no rotation (bit-identical to a rotation by 0)
// LOCATED: rec already is the record of the stage's segment (seg_locate done by the caller: the skewed stage loop of k_rk4)
template <bool AMP, int W, typename TabPtr, class HOOK = NoHook, bool ROT0 = false, bool LOCATED = false>
DEVINL void global_base(TabPtr tab, const GeoacDevParams& P, int& seg, double* rec, const double* y, double sth0, double cth0, double dlat, double* dy, GlobalStage& S, const HOOK& hook = HOOK(), double* rcp0 = nullptr, bool first = false){
    const double r = y[0];
    const double n0 = y[3], n1 = y[4], n2 = y[5];
    const double xe = clampq(r, P.x_min, P.x_max);
    double sth, cth;
    if(ROT0){ sth = sth0; cth = cth0; }
    else rot_small(sth0, cth0, dlat, sth, cth);                  // sin / cos of the stage latitude from the step's (stage 0: zero angle, exact identity)
    // |nu|   (Global.cpp:249)
    const double nn  = __builtin_fma(n0, n0, __builtin_fma(n1, n1, n2 * n2));
    const double inm0 = frsq(nn);
#if GEOAC_RCPC
    // 1/r and 1/cos(lat) of stages 1-3 by ONE third-order Newton step from the step's stage-0 values (r and cos(lat) move by < 1e-5 of themselves within a step: e^3 < 1e-15),
    // not from a fresh transcendental seed (17 issue cycles each against 4.4 for a multiply-add)
    double ir, ico;
    if(ROT0 || first){ ir = frcp(r); ico = frcp(cth); rcp0[0] = ir; rcp0[1] = ico; }     // (first: k_rk4_duo rolls all four stages into one loop - stage 0 by a run-time flag)
    else {
        const double e1 = __builtin_fma(-r, rcp0[0], 1.0), e2 = __builtin_fma(-cth, rcp0[1], 1.0);
        ir = __builtin_fma(rcp0[0], __builtin_fma(e1, e1, e1), rcp0[0]);
        ico = __builtin_fma(rcp0[1], __builtin_fma(e2, e2, e2), rcp0[1]);
    }
#else
    const double ir  = frcp(r);
    const double ico = frcp(cth);
#endif
    Atm9 a;
    if(LOCATED) seg_eval_at(rec, xe, a); else seg_cached_eval<W>(tab, P, xe, seg, rec, a);
    hook();
    const double inm = inm0, numag = nn * inm;
    const double u = a.u, v = a.v, du = a.du, dv = a.dv;

    // c = sqrt(gamR T), c' = gamR/(2c) T'                      (G2S_GlobalSpline1D.cpp:345-356)
    const double qT  = kGamR * a.T;
    const double ic  = frsq(qT);
    const double c   = qT * ic;
    const double hc  = (0.5 * kGamR) * ic;
    const double dc  = hc * a.dT;
    // group velocity c_g = c nu/|nu| + (0, v, u), |c_g|   (Global.cpp:250-255)
    const double cn  = c * inm;
    const double cg0 = cn * n0;
    const double cg1 = __builtin_fma(cn, n1, v);
    const double cg2 = __builtin_fma(cn, n2, u);
    const double icg = frsq(__builtin_fma(cg0, cg0, __builtin_fma(cg1, cg1, cg2 * cg2)));
    const double tn  = sth * ico;
    const double u0 = cg0 * icg, u1 = cg1 * icg, u2 = cg2 * icg;       // unit group-velocity vector
    const double G1 = ir, G2 = ir * ico;                               // GeoCoeff (Global.cpp:258-260)
    // GeoTerms (Global.cpp:263-269) with c_g substituted; the wind terms that cancel analytically are dropped:
    //   T1 = nu_r v - nu_r c_g1 + nu_p c_g2 tan      = -c/|nu| nu_r nu_t + nu_p c_g2 tan
    //   T2 = (..u..)cos + (..)sin - c_g2 (nu_r cos + nu_t sin) = -nu_p (v sin + c/|nu| (nu_r cos + nu_t sin))
    const double nc12 = __builtin_fma(n1, cg1, n2 * cg2);
    const double ncs  = __builtin_fma(n0, cth, n1 * sth);
    const double n2cg2 = n2 * cg2;
    const double T1 = __builtin_fma(n2cg2, tn, -cg0 * n1);
    const double T2 = -n2 * __builtin_fma(cn, ncs, v * sth);
    const double H0 = __builtin_fma(numag, dc, __builtin_fma(n1, dv, n2 * du));
    const double g1i = G1 * icg, g2i = G2 * icg;

    dy[0] = u0;
    dy[1] = G1 * u1;
    dy[2] = G2 * u2;
    dy[3] = -icg * __builtin_fma(ir, nc12, H0);                  // H0 + T0, T0 = nc12 / r
    dy[4] = -g1i * T1;
    dy[5] = -g2i * T2;

    if(AMP){
        const double ddc = __builtin_fma(hc, a.ddT, -(dc * dc) * ic);     // c'' = gamR/(2c) T'' - c'^2/c
        S.K2 = __builtin_fma(numag, ddc, __builtin_fma(n1, a.ddv, n2 * a.ddu));
        S.n0 = n0; S.n1 = n1; S.n2 = n2; S.inm = inm; S.cn = cn; S.icg = icg; S.dc = dc; S.du = du; S.dv = dv; S.v = v; S.cg2 = cg2;
        S.ir = ir; S.ico = ico; S.sth = sth; S.cth = cth; S.H0 = H0;
    }
}

// the products of the stage values that both derivative systems share (recomputed from the message in k_rk4_duo; in the one-wave kernels the
// compiler finds them among the base ray's own terms: the same operations on the same operands either way)
struct GlobalDerived { double cg0, cg1, u0, u1, u2, tn, G2, nc12, ncs, n2cg2, T1, T2, ir2, ico2, cnn1, cnn2; };
DEVINL void global_derive(const GlobalStage& S, GlobalDerived& D){
    D.cg0 = S.cn * S.n0;
    D.cg1 = __builtin_fma(S.cn, S.n1, S.v);
    D.u0 = D.cg0 * S.icg; D.u1 = D.cg1 * S.icg; D.u2 = S.cg2 * S.icg;
    D.tn = S.sth * S.ico;
    D.G2 = S.ir * S.ico;
    D.nc12 = __builtin_fma(S.n1, D.cg1, S.n2 * S.cg2);
    D.ncs  = __builtin_fma(S.n0, S.cth, S.n1 * S.sth);
    D.n2cg2 = S.n2 * S.cg2;
    D.T1 = __builtin_fma(D.n2cg2, D.tn, -D.cg0 * S.n1);
    D.T2 = -S.n2 * __builtin_fma(S.cn, D.ncs, S.v * S.sth);
    D.ir2 = S.ir * S.ir; D.ico2 = S.ico * S.ico;
    D.cnn1 = S.cn * S.n1; D.cnn2 = S.cn * S.n2;
}

// One launch-angle derivative system (Global.cpp:273-367 is this code written twice, once per angle).  ya: R_r, R_t, R_p, mu_r, mu_t, mu_p.
DEVINL void global_aux(const GlobalStage& S, const GlobalDerived& D, const double* ya, double* dya){
    const double n0 = S.n0, n1 = S.n1, n2 = S.n2, inm = S.inm, cn = S.cn, icg = S.icg, dc = S.dc, du = S.du, dv = S.dv, v = S.v, cg2 = S.cg2;
    const double ir = S.ir, sth = S.sth, cth = S.cth, H0 = S.H0, K2 = S.K2, G1 = S.ir;
    const double R0 = ya[0], R1 = ya[1];
    const double m0 = ya[3], m1 = ya[4], m2 = ya[5];
    const double dnu = __builtin_fma(n0, m0, __builtin_fma(n1, m1, n2 * m2)) * inm;       // d|nu|
    const double dca = R0 * dc, dva = R0 * dv;
    const double al  = inm * __builtin_fma(-cn, dnu, dca);                               // d(c/|nu|)
    const double a1  = __builtin_fma(n1, al, cn * m1);
    const double a2  = __builtin_fma(n2, al, cn * m2);
    const double dcg0 = __builtin_fma(n0, al, cn * m0);
    const double dcg1 = __builtin_fma(R0, dv, a1);
    const double dcg2 = __builtin_fma(R0, du, a2);
    const double e   = icg * __builtin_fma(D.u0, dcg0, __builtin_fma(D.u1, dcg1, D.u2 * dcg2)); // d|c_g| / |c_g|
    const double w0 = __builtin_fma(icg, dcg0, -D.u0 * e);
    const double w1 = __builtin_fma(icg, dcg1, -D.u1 * e);
    const double w2 = __builtin_fma(icg, dcg2, -D.u2 * e);
    const double dG1 = -R0 * D.ir2;
    const double dG2 = D.G2 * __builtin_fma(D.tn, R1, -R0 * ir);
    const double s22 = __builtin_fma(m2, cg2, n2 * dcg2);
    const double dT0 = __builtin_fma(dG1, D.nc12, ir * __builtin_fma(m1, D.cg1, __builtin_fma(n1, dcg1, s22)));
    const double dT1 = __builtin_fma(-D.cnn1, m0, __builtin_fma(-n0, a1, __builtin_fma(D.tn, s22, (D.n2cg2 * R1) * D.ico2)));
    const double dncs = __builtin_fma(m0, cth, __builtin_fma(m1, sth, R1 * __builtin_fma(n1, cth, -n0 * sth)));
    const double dT2 = -__builtin_fma(__builtin_fma(m2, v, n2 * dva), sth,
                         __builtin_fma((n2 * v) * R1, cth, __builtin_fma(a2, D.ncs, D.cnn2 * dncs)));

    dya[0] = w0;
    dya[1] = __builtin_fma(dG1, D.u1, G1 * w1);
    dya[2] = __builtin_fma(dG2, D.u2, D.G2 * w2);
    dya[3] = icg * (__builtin_fma(e, H0, -dT0)
                    - __builtin_fma(dnu, dc, __builtin_fma(m1, dv, __builtin_fma(m2, du, R0 * K2))));
    dya[4] = -icg * __builtin_fma(dG1, D.T1, G1 * dT1);
    dya[5] = -icg * __builtin_fma(dG2, D.T2, D.G2 * dT2);
}

// NQ = number of launch-angle derivative systems carried in y after the 6 base components: 2 = the reference layout
// (theta then phi), 1 = the two-lanes-per-ray kernel where each lane of a pair carries the base ray and ONE of the two systems.
// (Measured and dropped, round 4: locating the NEXT stage's spline segment here, under the launch-angle systems - the abscissa r0 + w dy[0] is known once the base
//  ray's slopes are - so that a dense fan of steep rays, where some lane of a wave changes segment in nearly every stage, finds its record in hand at the top of the
//  next stage.  Same bits, but 40 instructions per step: config 3 355-357 -> 366-368 ms per pass, metric pass 117 -> 122 ms; profiles/r04_b_cfg3_ab.txt.)
template <bool AMP, int NQ, bool ROT0 = false, typename TabPtr>
DEVINL void global_rhs(TabPtr tab, const GeoacDevParams& P, int& seg, double* rec, const double* y, double sth0, double cth0, double dlat, double* dy, double* rcp0 = nullptr){
    GlobalStage S;
    global_base<AMP, GEOAC_SEGW, TabPtr, NoHook, ROT0>(tab, P, seg, rec, y, sth0, cth0, dlat, dy, S, NoHook(), rcp0);
    if(AMP){
        GlobalDerived D;
        global_derive(S, D);
        #pragma unroll
        for(int q = 0; q < NQ; q++) global_aux(S, D, y + 6 + 6 * q, dy + 6 + 6 * q);
    }
}

// ------------------------------------------------------------------------------------------------
// rare-path (once per leg) helpers of the Global set; medium values from the global-memory tables
// ------------------------------------------------------------------------------------------------
struct Medium { double c, dc, u, du, v, dv, rho; };
DEVINL Medium medium_at(const GeoacDevParams& P, double x){
    double xe = clampd(x, P.x_min, P.x_max);
    int k = seg_guess(P.seg, P, xe);
    Atm9 a; seg_eval<false>(P.seg, k, xe, a);
    Medium m;
    m.c = sqrt(kGamR * a.T); m.dc = kGamR / (2.0 * m.c) * a.dT;
    m.u = a.u; m.du = a.du; m.v = a.v; m.dv = a.dv;
    m.rho = rho_eval(P, k, xe);
    return m;
}

// GeoAc_Jacobian: EquationSets.Global.cpp:594-607  (1/(r sin(lat)) in dp_ds: Q3)
DEVINL double global_jacobian(const Medium& m, const double* y){
    double r = y[0], th = y[1];
    double nu_mag = sqrt(y[3] * y[3] + y[4] * y[4] + y[5] * y[5]);
    double cp0 = m.c * y[3] / nu_mag, cp1 = m.c * y[4] / nu_mag + m.v, cp2 = m.c * y[5] / nu_mag + m.u;
    double cpm = sqrt(cp0 * cp0 + cp1 * cp1 + cp2 * cp2);
    double dr_ds = cp0 / cpm, dt_ds = 1.0 / r * cp1 / cpm, dp_ds = 1.0 / (r * sin(th)) * cp2 / cpm;
    double dr_dlt = y[6], dt_dlt = y[7], dp_dlt = y[8];
    double dr_dlp = y[12], dt_dlp = y[13], dp_dlp = y[14];
    return r * r * cos(th) * (dr_ds * (dt_dlt * dp_dlp - dt_dlp * dp_dlt) - dr_dlt * (dt_ds * dp_dlp - dp_ds * dt_dlp)
                              + dr_dlp * (dt_ds * dp_dlt - dp_ds * dt_dlt));
}

// GeoAc_Amplitude: EquationSets.Global.cpp:610-629 (c_prop0[1..2] over the ARRIVAL nu_mag: Q4)
DEVINL double global_amplitude(const GeoacDevParams& P, const Medium& m, const Medium& m0, const double* y,
                               double c0, double nu_mag0, double th_l, double ph_l, double D){
    double nu0v[3] = { sin(th_l), cos(th_l) * sin(ph_l), cos(th_l) * cos(ph_l) };
    double nu_mag = (c0 - y[4] * m.v - y[5] * m.u) / m.c;
    double cp0 = m.c * y[3] / nu_mag, cp1 = m.c * y[4] / nu_mag + m.v, cp2 = m.c * y[5] / nu_mag + m.u;
    double cq0 = c0 * nu0v[0] / nu_mag0, cq1 = c0 * nu0v[1] / nu_mag + m0.v, cq2 = c0 * nu0v[2] / nu_mag + m0.u;
    double cpm = sqrt(cp0 * cp0 + cp1 * cp1 + cp2 * cp2);
    double cqm = sqrt(cq0 * cq0 + cq1 * cq1 + cq2 * cq2);
    double num = m.rho * nu_mag * (m.c * m.c * m.c) * cqm * cos(th_l);
    double den = m0.rho * nu_mag0 * (m0.c * m0.c * m0.c) * cpm * D;
    return 1.0 / (4.0 * kPi) * sqrt(fabs(num / den));
}

// ------------------------------------------------------------------------------------------------
// SuthBass_Alpha (Atmo_State.Absorption.Global.cpp:12-141 / Atmo_State.Absorption.cpp:14-143); zr = altitude above sea level
// ------------------------------------------------------------------------------------------------
// square root of a non-negative number through the rsq seed (<= 1 ulp, not correctly rounded; 0 -> 0)
DEVINL double fsqrt(double x){ double r = x * frsq(x); return (x > 0.0) ? r : 0.0; }

// The ~35 divisions and most square roots of the reference expression go through frcp / frsq (<= 1 ulp each): this kernel runs at
// half the FP64 vector peak, and an IEEE division costs three times an frcp.  Kept correctly rounded: sqrt(1 + nu^2), whose
// difference to 1 the classical term takes (catastrophic cancellation in the reference itself - the last bit of that root is
// worth 1e-6 of a_cl at 80 km).
// PARTS (k_atab_build): also returns the three smooth functions the absorption table holds - see atab_eval.
template <bool PARTS = false>
DEVINL double suthbass_alpha(const GeoacDevParams& P, double zr, double c_snd, double rho, double freq, double T_o, double P_o, double cbrt_To, double* parts = nullptr){
    const double mu_o = 18.192E-6, S = 117.0;
    double cm = c_snd * 1000.0;
    double T_z = cm * cm * (1.0 / (kRgas * kGam));
    double P_z = rho * (cm * cm) * (1000.0 / kGam);
    const double iTz = frcp(T_z), iTo = frcp(T_o);
    double mu = mu_o * fsqrt(T_z * iTo) * ((1.0 + S * iTo) * frcp(1.0 + S * iTz));
    double nu = (8.0 * kPi * freq * mu) * frcp(3.0 * P_z);

    double z2 = zr * zr, z3 = z2 * zr, z4 = z2 * z2, z5 = z4 * zr;
    double X0, X1, X2, X3, X4, X5, X6;
    X0 = (zr > 90.) ? fexp10(49.296 - (1.5524 * zr) + (1.8714E-2 * z2) - (1.1069E-4 * z3) + (3.199E-7 * z4) - (3.6211E-10 * z5))
                    : P.sb_const[0];          // 10^-0.67887 (host pow)
    X1 = (zr > 76.) ? fexp10((1.3972E-1) - (5.6269E-3 * zr) + (3.9407E-5 * z2) - (1.0737E-7 * z3))
                    : P.sb_const[1];          // 10^-0.10744
    X2 = P.sb_const[2];                     // 10^-3.3979
    // one exponential per species: the branch of the reference's piecewise fits selects the exponent
    X3 = fexp10((zr > 80.) ? -4.234 - (3.0975E-2 * zr)
                           : -19.027 + (1.3093 * zr) - (4.6496E-2 * z2) + (7.8543E-4 * z3) - (6.5169E-6 * z4) + (2.1343E-8 * z5));
    X4 = fexp10((zr > 95.) ? -3.2456 + (4.6642E-2 * zr) - (2.6894E-4 * z2) + (5.264E-7 * z3)
                           : -11.195 + (1.5408E-1 * zr) - (1.4348E-3 * z2) + (1.0166E-5 * z3));
    X5 = fexp10(-53.746 + (1.5439 * zr) - (1.8824E-2 * z2) + (1.1587E-4 * z3) - (3.5399E-7 * z4) + (4.2609E-10 * z5));
    X6 = fexp10((zr > 30.) ? -4.2563 + (7.6245E-2 * zr) - (2.1824E-3 * z2) - (2.3010E-6 * z3) + (2.4265E-7 * z4) - (1.2500E-09 * z5)
                           : -1.7491 + (4.4986E-2 * zr) - (6.8549E-2 * z2) + (5.4639E-3 * z3) - (1.5539E-4 * z4) + (1.5063E-06 * z5));
    double X_ON = (X0 + X1) * (1.0 / 0.9903);

    double rc = rcbrt(T_z);                                  // T_z^(-1/3)
    double Zr0 = 54.1 * fexp(-17.3 * rc);
    double Zr1 = 63.3 * fexp(-16.7 * rc);
    double Z_rot_ = (Zr0 * Zr1) * frcp(__builtin_fma(X1, Zr0, X0 * Zr1));      // 1 / (X1/Zr1 + X0/Zr0)

    const double sigma = P.sb_const[3];                      // 5/sqrt(21)
    double nn = (4.0 / 5.0) * P.sb_const[4] * Z_rot_;         // sqrt(3/7)
    double chi = 3.0 * nn * nu * 0.25;
    double cchi = 2.36 * chi;

    const double ic = frcp(c_snd);
    double w0 = 2.0 * kPi * freq * ic;
    double nu2 = nu * nu, sq = sqrt(1.0 + nu2);              // IEEE: see above
    const double cc2 = cchi * cchi, sc2 = (sigma * cchi) * (sigma * cchi);
    double a_cl  = w0 * fsqrt(0.5 * (sq - 1.0) * (1.0 + cc2) * frcp((1.0 + nu2) * (1.0 + sc2)));
    double a_rot = w0 * X_ON * ((sigma * sigma - 1.0) * chi * frcp(2 * sigma)) * fsqrt(0.5 * (sq + 1.0) * frcp((1.0 + nu2) * (1.0 + cc2)));
    double a_diff = 0.003 * a_cl;

    double Tr = __builtin_fma(cbrt_To, rc, -1.0);            // (T_z/T_o)^(-1/3) - 1 = cbrt(T_o) T_z^(-1/3) - 1
    double A1 = (X0 + X1) * 24.0 * fexp(-9.16 * Tr);
    double A2 = (X4 + X5) * 2400.0;
    double B  = 40400.0 * fexp(10.0 * Tr);
    double C  = 0.02 * fexp(-11.2 * Tr);
    double D  = 0.391 * fexp(8.41 * Tr);
    double Ee = 9.0 * fexp(-19.9 * Tr);
    double F  = 60000.0;
    double G  = 28000.0 * fexp(-4.17 * Tr);
    double H  = 22000.0 * fexp(-7.68 * Tr);
    double I  = 15100.0 * fexp(-10.4 * Tr);
    double J  = 11500.0 * fexp(-9.17 * Tr);
    double K  = (8.48E08) * fexp(9.17 * Tr);
    double L  = fexp(-7.72 * Tr);
    double ZZ = H * X2 + I * (X0 + 0.5 * X4) + J * (X1 + 0.5 * X5) + K * (X6 + X3);
    double hu = 100.0 * (X3 + X6);
    double pm = (P_z * mu_o) * frcp(P_o * mu);
    double fv[4] = { pm * (A1 + A2 + B * hu * (C + hu) * (D + hu)), pm * (Ee + F * X3 + G * X6), pm * ZZ, pm * (1.2E5) * L };
    const double Theta[4] = { 2239.1, 3352.0, 915.0, 1037.0 };
    const double Cp_R[4] = { 3.5, 3.5, 4.0, 4.0 }, Cv_R[4] = { 2.5, 2.5, 3.0, 3.0 };
    double Xm[4] = { X0, X1, X2, X3 };
    double a_vib = 0.0;
    const double f2x2 = 2 * (freq * freq);
    #pragma unroll
    for(int m = 0; m < 4; m++){
        double q = Theta[m] * iTz;
        double ex = fexp(-q);
        double om = 1 - ex;
        double C_R = ((q * q) * ex) * frcp(om * om);
        double ifv = frcp(fv[m]);
        double fr = freq * ifv;
        // (A_max / c) (2 f^2 / f_vib) / (1 + (f/f_vib)^2),  A_max = X (pi/2) C_R / (Cp (Cv + C_R))
        a_vib += (Xm[m] * (kPi / 2) * C_R) * (ic * f2x2 * ifv) * frcp((Cp_R[m] * (Cv_R[m] + C_R)) * __builtin_fma(fr, fr, 1.0));
    }
    if(PARTS){
        // alpha = S + sqrt((sqrt(1 + N) - 1) Q):  S = the rotational and vibrational terms, N = nu^2, Q = what multiplies (sq - 1) under the root
        // of the classical (+ diffusion, 0.003 of it) term, constants folded in
        const double kk = P.tweak_abs * 8.685889;
        parts[0] = (a_rot + a_vib) * kk;
        parts[1] = nu2;
        parts[2] = (w0 * w0) * 0.5 * (1.0 + cc2) * frcp((1.0 + nu2) * (1.0 + sc2)) * ((1.003 * kk) * (1.003 * kk));
    }
    return (a_cl + a_rot + a_diff + a_vib) * P.tweak_abs * 8.685889;
}

// ------------------------------------------------------------------------------------------------
// Absorption table of the stratified sets.  There c and rho are functions of the height coordinate alone, so SuthBass_Alpha - which the
// reference evaluates at the midpoint of every path segment (Global.cpp:634-670 and twins; ~1 500 FP64 instructions per 64 midpoints
// here) - is a function of one variable.  It is not a SMOOTH function of it as the reference (and suthbass_alpha above) computes it: the
// classical term takes sqrt(1 + nu^2) - 1 with nu^2 between 1e-20 and 1e-7, so below ~70 km it is a staircase in the last bits of
// that root (exactly 0 below ~35 km), worth up to 2e-5 of alpha - and parity means reproducing the staircase.  So the table holds,
// per spline segment, the degree-5 interpolants (Chebyshev nodes, kept as polynomials in s = 2 t / h - 1) of the three smooth pieces
//     S = (a_rot + a_vib) k,   N = nu^2,   Q = (w0^2 / 2) (1 + cchi^2) / ((1 + nu^2)(1 + (sigma cchi)^2)) (1.003 k)^2,   k = tweak x 8.685889
// and the post-pass forms alpha = S + sqrt((sqrt(1 + N) - 1) Q) with the same correctly rounded root: N is reproduced to ~1e-13 relative,
// far inside the spacing of the doubles next to 1, so 1 + N rounds as in the exact routine.  k_atab_build samples the pieces from the
// exact routine itself (inside a spline segment the medium is a cubic and the pieces analytic) and checks the reassembled alpha against
// it at eight other points of the segment: an entry that misses 1e-10 relative anywhere (a branch point of the reference's piecewise
// fits inside the segment, a very long segment) is flagged, and the post-pass evaluates the segments that fall into it exactly (fix-up
// pass of k_postpass).  Two more entries cover the strips just below the first and above the last node, where the reference clamps
// the medium but not the height (the row below the ground, the row above the top).
// ------------------------------------------------------------------------------------------------
DEVINL double atab_poly(const double* c, double s){              // degree 5
    double p = c[5];
    p = __builtin_fma(p, s, c[4]); p = __builtin_fma(p, s, c[3]); p = __builtin_fma(p, s, c[2]);
    p = __builtin_fma(p, s, c[1]); p = __builtin_fma(p, s, c[0]);
    return p;
}
// e: a table entry (in memory or in registers): [0] 2 / h (NEGATIVE: the entry is flagged), [1..6] S, [7..12] N, [13..18] Q; t: offset from its left end
DEVINL double atab_eval(const double* e, double t){
    const double s = __builtin_fma(t, fabs(e[0]), -1.0);
    const double S = atab_poly(e + 1, s), N = atab_poly(e + 7, s), Q = atab_poly(e + 13, s);
    const double sq = sqrt(1.0 + N);                              // IEEE, as in suthbass_alpha
    return S + fsqrt((sq - 1.0) * Q);
}
// the same with the entry in LDS, coefficient c at e[c * 256] (k_postpass_tab<.., TBL>): one piece after the other, so that six coefficients are
// in registers at a time and not all nineteen (the scheduler would fetch them together otherwise: 38 registers the kernel does not have)
DEVINL double atab_poly_lds(const double* e, int c0, double s){
    double p = e[(c0 + 5) * 256];
    p = __builtin_fma(p, s, e[(c0 + 4) * 256]); p = __builtin_fma(p, s, e[(c0 + 3) * 256]); p = __builtin_fma(p, s, e[(c0 + 2) * 256]);
    p = __builtin_fma(p, s, e[(c0 + 1) * 256]); p = __builtin_fma(p, s, e[c0 * 256]);
    return p;
}
DEVINL double atab_eval_lds(const double* e, double e0, double t){
    const double s = __builtin_fma(t, fabs(e0), -1.0);
    const double S = atab_poly_lds(e, 1, s);
    __builtin_amdgcn_sched_barrier(0);
    const double N = atab_poly_lds(e, 7, s);
    __builtin_amdgcn_sched_barrier(0);
    const double Q = atab_poly_lds(e, 13, s);
    __builtin_amdgcn_sched_barrier(0);
    const double sq = sqrt(1.0 + N);                              // IEEE, as in suthbass_alpha
    return S + fsqrt((sq - 1.0) * Q);
}
// table entry and offset within it of abscissa x; xe = x clamped to the profile, k / x0 = spline segment of xe and its left node.  t < 0
// or t > the entry's length: beyond the strips, not served
DEVINL int atab_locate(const GeoacDevParams& P, double x, double xe, int k, double x0, double& t, bool& out){
    const bool below = x < P.x_min, above = x > P.x_max;
    t = below ? x - P.atab_lo : (above ? x - P.x_max : xe - x0);        // (atab_lo = x_min - atab_D, formed on the host)
    out = (t < 0.0) | (above & (t > P.atab_D));
    return below ? P.nseg : (above ? P.nseg + 1 : k);
}
// alpha [dB/km, tweak included] at abscissa x (see atab_locate); -1: the table does not serve this point
DEVINL double atab_alpha(const GeoacDevParams& P, double x, double xe, int k, double x0){
    double t; bool out;
    const int e = atab_locate(P, x, xe, k, x0, t, out);
    const double* r = P.atab + (size_t)e * GEOAC_ATABW;
    const double a = atab_eval(r, t);
    return (out | (r[0] < 0.0)) ? -1.0 : a;
}
// segment index of xe: first guess by multiplication, then the walk over the node abscissae (seg_find)
template <typename TabPtr>
DEVINL int seg_guess_mul(TabPtr tab, const GeoacDevParams& P, double xe){
    return seg_find(tab, P.nseg, xe, (int)((xe - P.x_min) * P.seg_per_x));
}
// a path segment as k_postpass_tab sees it: abscissa of the midpoint, path lengths of the travel-time and the attenuation integral, and up to
// four set-specific values of the midpoint (Global: nu and 1 / |nu|; 3-D: nu_z)
struct PPGeom { double x, ds_tt, ds_at, a0, a1, a2, a3; };
// exact post-pass of one segment of a stratified set: geometry, medium at the midpoint, travel time, SuthBass_Alpha x path length
template <class EQ>
DEVINL void pp_exact(const GeoacDevParams& P, const double* aux, const double* A, const double* B, double& tt, double& at);
// T, u, v of the segment record p at abscissa x (values only)
DEVINL void seg_eval_f(const double* __restrict__ p, double x, double& T, double& u, double& v){
    const double t = x - p[0], t6 = t * (1.0 / 6.0);
    const double fT = __builtin_fma(t, p[5], p[4]), fu = __builtin_fma(t, p[9], p[8]), fv = __builtin_fma(t, p[13], p[12]);
    T = __builtin_fma(t, __builtin_fma(t6, p[4] + (p[4] + fT), p[3]), p[2]);
    u = __builtin_fma(t, __builtin_fma(t6, p[8] + (p[8] + fu), p[7]), p[6]);
    v = __builtin_fma(t, __builtin_fma(t6, p[12] + (p[12] + fv), p[11]), p[10]);
}


// ------------------------------------------------------------------------------------------------
// 3-D stratified Cartesian set: fused GeoAc_UpdateSources + GeoAc_EvalSrcEq (EquationSets.3DStratified.cpp:203-310)
// y: x, y, z, nu_z | X_th, Y_th, Z_th, mu_z_th | X_ph, Y_ph, Z_ph, mu_z_ph ; nu_x, nu_y and their launch-angle
// derivatives are constants of the ray (stratified medium) kept in RayCtx::a[0..5]
// ------------------------------------------------------------------------------------------------
struct RayCtx {
    double c0;        // Global/3D: sound speed at the source; 2D: effective sound speed at the source
    double nu0;       // Global: 1/MachScalar
    double a[6];      // Global: sin/cos(lat), sin/cos(lon - lon_src) carried along the ray
                      // 3D: nu_x, nu_y, mu_x_th, mu_y_th, mu_x_ph, mu_y_ph ; 2D: cos(phi), sin(phi), cos(theta), sin(theta)
    double t[4];      // Global: proposed sin/cos for the row under test
    double cur[4];    // Global (stratified): sin/cos(lat), sin/cos(lon - lon_src) of the current row - a[] then holds the REFERENCE point they are
                      // rotated from: lat_ref, sin, cos, (lon - lon_src)_ref, sin, cos (EqGlobal::checks)
    mutable double rcp0[2];           // stratified Global set (GEOAC_RCPC): 1/r, 1/cos(lat) of the step's stage 0
    mutable double rec[GEOAC_SEGW];   // 1-D sets: the spline record of the segment the ray is in (x0, x1, cubics of T, u, v), seg_cached_eval
    mutable int ckey; // record-cache kernels: (segment, node) key of the records this lane holds in LDS (-1: none)
    mutable int kxy;  // grid sets: horizontal cell of the previous evaluation, kx << 16 | ky (-1: none), grid_locate's hint
    mutable double cell[4];   // and its node coordinates X1, X2, Y1, Y2
};

template <bool AMP, int NQ = 2, typename TabPtr = const double*>
DEVINL void cart3_rhs(TabPtr tab, const GeoacDevParams& P, int& seg, const RayCtx& C, const double* y, double* dy, int q0 = 0){
    const double nz = y[3];
    const double nx = C.a[0], ny = C.a[1];
    const double xe = clampq(y[2], P.x_min, P.x_max);
    Atm9 a;
    seg_cached_eval(tab, P, xe, seg, C.rec, a);
    const double u = a.u, v = a.v, du = a.du, dv = a.dv;
    const double qT = kGamR * a.T;
    const double ic = frsq(qT);
    const double c  = qT * ic;
    const double hc = (0.5 * kGamR) * ic;
    const double dc = hc * a.dT;
    const double numag = (C.c0 - nx * u - ny * v) * ic;            // c0/c (1 - (nu.wind)/c0)     (3DStratified.cpp:214)
    const double inm = frcp(numag);
    const double cn  = c * inm;
    const double cp0 = __builtin_fma(cn, nx, u), cp1 = __builtin_fma(cn, ny, v), cp2 = cn * nz;
    const double icp = frsq(__builtin_fma(cp0, cp0, __builtin_fma(cp1, cp1, cp2 * cp2)));
    const double u0 = cp0 * icp, u1 = cp1 * icp, u2 = cp2 * icp;
    const double H = __builtin_fma(numag, dc, __builtin_fma(nx, du, ny * dv));
    dy[0] = u0; dy[1] = u1; dy[2] = u2;
    dy[3] = -icp * H;
    if(AMP){
        const double ddc = __builtin_fma(hc, a.ddT, -(dc * dc) * ic);
        const double K2  = __builtin_fma(numag, ddc, __builtin_fma(nx, a.ddu, ny * a.ddv));
        #pragma unroll
        for(int q = 0; q < NQ; q++){
            // NQ = 1 (two-lanes-per-ray kernel): this lane carries system q0 in y[4..7]; its constant slowness derivatives sit in C.a[2 + 2 q0 ..]
            const double mx = (NQ == 1) ? (q0 ? C.a[4] : C.a[2]) : C.a[2 + 2 * q], my = (NQ == 1) ? (q0 ? C.a[5] : C.a[3]) : C.a[3 + 2 * q];
            const double mz = y[7 + 4 * q], Za = y[6 + 4 * q];
            const double dnu = __builtin_fma(nx, mx, __builtin_fma(ny, my, nz * mz)) * inm;
            const double al  = inm * __builtin_fma(dc, Za, -cn * dnu);
            const double dcp0 = __builtin_fma(nx, al, __builtin_fma(cn, mx, du * Za));
            const double dcp1 = __builtin_fma(ny, al, __builtin_fma(cn, my, dv * Za));
            const double dcp2 = __builtin_fma(nz, al, cn * mz);
            const double e = icp * __builtin_fma(u0, dcp0, __builtin_fma(u1, dcp1, u2 * dcp2));
            dy[4 + 4 * q] = __builtin_fma(icp, dcp0, -u0 * e);
            dy[5 + 4 * q] = __builtin_fma(icp, dcp1, -u1 * e);
            dy[6 + 4 * q] = __builtin_fma(icp, dcp2, -u2 * e);
            dy[7 + 4 * q] = icp * (e * H - __builtin_fma(dnu, dc, __builtin_fma(mx, du, __builtin_fma(my, dv, K2 * Za))));
        }
    }
}

// 2-D effective-sound-speed set (EquationSets.2DStratified.cpp:135-181).  y: r, z, nu_z | R, Z, mu_z
template <bool AMP, typename TabPtr>
DEVINL void cart2_rhs(TabPtr tab, const GeoacDevParams& P, int& seg, const RayCtx& C, const double* y, double* dy){
    const double nz = y[2];
    const double cph = C.a[0], sph = C.a[1], cth = C.a[2], sth = C.a[3];
    const double xe = clampq(y[1], P.x_min, P.x_max);
    Atm9 a;
    seg_cached_eval(tab, P, xe, seg, C.rec, a);
    const double qT = kGamR * a.T;
    const double ic = frsq(qT);
    const double c  = qT * ic;
    const double hc = (0.5 * kGamR) * ic;
    const double dc = hc * a.dT;
    const double ce  = c + a.u * cph + a.v * sph;
    const double dce = dc + a.du * cph + a.dv * sph;
    const double ic0 = C.nu0;                                        // 1 / c_eff_0
    const double ice = frcp(ce);
    const double g = ce * ic0;
    dy[0] = g * cth;
    dy[1] = g * nz;
    dy[2] = -C.c0 * ice * ice * dce;
    if(AMP){
        const double ddc  = __builtin_fma(hc, a.ddT, -(dc * dc) * ic);
        const double ddce = ddc + a.ddu * cph + a.ddv * sph;
        const double Zt = y[4], mz = y[5];
        const double h = dce * Zt * ic0;
        const double q = dce * ice;
        dy[3] = h * cth - g * sth;
        dy[4] = h * nz + g * mz;
        dy[5] = (2 * (q * q) - ddce * ice) * (C.c0 * ice) * Zt;
    }
}

// ================================================================================================
// Equation-set policies: everything the generic kernels need to know about a set
// ================================================================================================
template <bool AMP_> struct EqGlobal {
    static constexpr bool AMP = AMP_;
    static constexpr bool SEG1D = true;                             // the ray keeps the spline record of its segment in registers (RayCtx::rec)
    static constexpr int AUX_SAVE = 6;                              // entries of RayCtx::a the RK4 kernel changes (the reference point of the carried sin / cos)
    static constexpr bool COOP = false; static constexpr bool CACHE = false; static constexpr bool LDS_STATE = false; static constexpr int XCHG_BYTES = 0; static constexpr bool PP_TILE = false; static constexpr bool PP_DEDUP = false; static constexpr int SYS_SHIFT = 0; static constexpr bool ROW_SPLIT = false;
    static constexpr int PP_WAVES = 3;                              // post-pass waves per SIMD (168 registers; at 128 it spills 148 B and runs 1.5 x slower)
    static constexpr int E = AMP_ ? 18 : 6, PW = 6, HIDX = 0, LANES = 1;
    static constexpr bool SPLIT = false;                            // true: the lanes of a ray carry DIFFERENT parts of its state (EqGlobalPair)
    static constexpr int NB = 6, NS = 6;                            // base-ray components, components per launch-angle derivative system
    static constexpr bool KM2 = false, KM2_MEM = false, HMAX_PER_LEG = false;        // linear intercept only (Q1); turning height accumulates over legs

    // GeoAc_SetInitialConditions: EquationSets.Global.cpp:76-136
    static DEVINL void init(const GeoacDevParams& P, double th, double ph, double* y, RayCtx& C){
        double z_src = P.src[0] < P.z_grnd ? P.z_grnd : P.src[0];     // GeoAcGlobal_main.cpp:165
        double r0 = z_src + P.r_earth;
        double lat0 = P.src[1] * kPi / 180.0, lon0 = P.src[2] * kPi / 180.0;
        Medium m = medium_at(P, r0);
        double c0 = m.c;
        double Mach[3] = { 0.0, m.v / c0, m.u / c0 };
        double sth = sin(th), cth = cos(th), sph = sin(ph), cph = cos(ph);
        double nu0[3]  = { sth, cth * sph, cth * cph };
        double mlt[3]  = { cth, -sth * sph, -sth * cph };
        double mlp[3]  = { 0.0, cth * cph, -cth * sph };
        double MS = 1.0 + (nu0[0] * Mach[0] + nu0[1] * Mach[1] + nu0[2] * Mach[2]);
        y[0] = r0; y[1] = lat0; y[2] = lon0;
        for(int e = 0; e < 3; e++) y[3 + e] = nu0[e] / MS;
        if(AMP){
            double dlt = mlt[0] * Mach[0] + mlt[1] * Mach[1] + mlt[2] * Mach[2];
            double dlp = mlp[0] * Mach[0] + mlp[1] * Mach[1] + mlp[2] * Mach[2];
            for(int e = 0; e < 3; e++){
                y[9 + e]  = mlt[e] / MS - nu0[e] / (MS * MS) * dlt;
                y[15 + e] = mlp[e] / MS - nu0[e] / (MS * MS) * dlp;
            }
        }
        C.c0 = c0; C.nu0 = 1.0 / MS;
        C.a[0] = lat0; fsincos(lat0, C.a[1], C.a[2]);
        C.a[3] = 0.0; C.a[4] = 0.0; C.a[5] = 1.0;
        resume(P, C, y);
    }
    static DEVINL void fan_init(const GeoacDevParams& P){}
    static DEVINL double height(const GeoacDevParams& P, const double* y){ return y[0] - P.r_earth; }
    static DEVINL double above_ground(const GeoacDevParams& P, const double* y){ return y[0] - P.ground; }

    // sin / cos of the latitude and of (lon - lon_src) are carried along the ray: rotated from a REFERENCE point (a[]: angle, sin, cos - exact
    // values, in the ray's state) by the angle to the row in question, which stays below GEOAC_ROT_MAX (1e-3 rad, ~6 km of travel: the series of
    // rot_small are then exact to 1e-17); a row further away becomes the new reference (fsincos, rare and per ray).  No error accumulates, and the
    // values of a row depend on the ray's own rows only - not on where an epoch or a kernel began.
    static DEVINL void resume(const GeoacDevParams& P, RayCtx& C, const double* y){   // current row from the reference (kernel entry)
        rot_small(C.a[1], C.a[2], y[1] - C.a[0], C.cur[0], C.cur[1]);
    }
    template <typename TabPtr>
    static DEVINL void rhs(TabPtr tab, const GeoacDevParams& P, int& seg, const RayCtx& C, const double* y0, const double* yt, int stage, double* dy){
        // (stage 0 - a constant where the step loop peels it: the stage latitude is the step's, no rotation)
        if(stage == 0) global_rhs<AMP, 2, true>(tab, P, seg, C.rec, yt, C.cur[0], C.cur[1], 0.0, dy, C.rcp0);
        else global_rhs<AMP, 2>(tab, P, seg, C.rec, yt, C.cur[0], C.cur[1], yt[1] - y0[1], dy, C.rcp0);
    }
    // GeoAc_BreakCheck / GeoAc_GroundCheck on the new row (Global.cpp:500-522)
    static DEVINL void checks(const GeoacDevParams& P, RayCtx& C, const double* y, const double* yn, int k, bool& brk, bool& gnd){
        const double lon0 = P.src[2] * kPi / 180.0;
        const double dl = yn[1] - C.a[0], pl = yn[2] - lon0;
        if(__builtin_expect(fabs(dl) > GEOAC_ROT_MAX, 0)){ C.a[0] = yn[1]; fsincos(yn[1], C.a[1], C.a[2]); C.t[0] = C.a[1]; C.t[1] = C.a[2]; }   // (rare, per ray)
        else rot_small(C.a[1], C.a[2], dl, C.t[0], C.t[1]);
        // haversine of the great-circle range: hav = sin^2(dlat/2) + cos(lat0) cos(lat) sin^2(dlon/2), with 2 sin^2(x/2) = 1 - cos x;
        // range = 2 R asin(sqrt(hav)) > limit  <=>  hav > sin^2(limit / 2R).
        // hav <= sin^2 a + sin^2 b <= sin^2(a + b) for a = |dlat| / 2, b = |dlon| / 2, a + b <= pi / 2 (the difference is 2 sin a sin b cos(a + b)):
        // while a + b stays below limit / 2R (less 1e-9 of it: P.range_skip) the test cannot fire, and the ~50 instructions of the longitude
        // rotation and the haversine are skipped by a wave whose rays are all that close to the source.  The longitude reference point goes
        // stale meanwhile; the first row that is tested finds it further than GEOAC_ROT_MAX away and takes a new one.
        bool far = false;
        if(fabs(yn[1] - P.src[1] * kPi / 180.0) * 0.5 + fabs(pl) * 0.5 >= P.range_skip){
            const double dp = pl - C.a[3];
            double tp0, tp1;
            if(__builtin_expect(fabs(dp) > GEOAC_ROT_MAX, 0)){ C.a[3] = pl; fsincos(pl, C.a[4], C.a[5]); tp0 = C.a[4]; tp1 = C.a[5]; }
            else rot_small(C.a[4], C.a[5], dp, tp0, tp1);
            const double sl0 = P.src_trig[0], cl0 = P.src_trig[1];
            double hav = __builtin_fma(cl0 * C.t[1], 0.5 * (1.0 - tp1), 0.5 * (1.0 - __builtin_fma(C.t[1], cl0, C.t[0] * sl0)));
            far = hav > P.range_thresh;
        }
        brk = (yn[0] > P.vert_limit) || far;
        gnd = yn[0] < P.ground;
    }
    static DEVINL void accept(RayCtx& C){ C.cur[0] = C.t[0]; C.cur[1] = C.t[1]; }
    static DEVINL void restart(const GeoacDevParams& P, RayCtx& C, const double* y){      // start of a leg: the reflected row is the new reference
        C.a[0] = y[1]; fsincos(y[1], C.a[1], C.a[2]);
        C.a[3] = y[2] - P.src[2] * kPi / 180.0; fsincos(C.a[3], C.a[4], C.a[5]);
        C.cur[0] = C.a[1]; C.cur[1] = C.a[2];
    }
    // arrival row: GeoAcGlobal_main.cpp:296-317
    static DEVINL void arrival(const GeoacDevParams& P, const RayCtx& C, int slot, const double* yn, double* R){
        const double lat0 = P.src[1] * kPi / 180.0, lon0 = P.src[2] * kPi / 180.0;
        Medium m = medium_at(P, yn[0]);
        double z_src = P.src[0] < P.z_grnd ? P.z_grnd : P.src[0];
        double incl = -asin(m.c / C.c0 * yn[3]) * 180.0 / kPi;
        double baz = 90.0 - atan2(-yn[4], -yn[5]) * 180.0 / kPi;
        if(baz < -180.0) baz += 360.0;
        if(baz > 180.0) baz -= 360.0;
        double g1 = sin((yn[1] - lat0) / 2.0); g1 *= g1;
        double g2 = sin((yn[2] - lon0) / 2.0); g2 = cos(lat0) * cos(yn[1]) * g2 * g2;
        R[GEOAC_REC_INCL] = incl;
        R[GEOAC_REC_BACKAZ] = baz;
        R[GEOAC_REC_RANGE] = 2.0 * P.r_earth * asin(sqrt(g1 + g2));
        if(AMP){
            double amp, D;
            amp_jac(P, C, slot, yn, amp, D);
            R[GEOAC_REC_AMP] = amp;
            R[GEOAC_REC_JACOB] = D;
        }
    }
    // GeoAc_Amplitude / GeoAc_Jacobian of an arbitrary row (arrival rows, raypath samples, caustic detection)
    static DEVINL void amp_jac(const GeoacDevParams& P, const RayCtx& C, int slot, const double* yr, double& amp, double& D){
        double z_src = P.src[0] < P.z_grnd ? P.z_grnd : P.src[0];
        Medium m = medium_at(P, yr[0]);
        Medium m0 = medium_at(P, z_src + P.r_earth);
        D = global_jacobian(m, yr);
        double th_l = P.theta_deg[slot] * kPi / 180.0, ph_l = kPi / 2.0 - P.phi_deg[slot] * kPi / 180.0;
        amp = global_amplitude(P, m, m0, yr, C.c0, C.nu0, th_l, ph_l, D);
    }
    // GeoAc_ApproximateIntercept + GeoAc_SetReflectionConditions (Global.cpp:140-205); Q1: the quadratic term is a
    // discarded expression in the reference -> linear intercept.  y = row k-1 in, new leg's row 0 out.
    static DEVINL void reflect(const GeoacDevParams& P, const RayCtx& C, const double* yn, double* y, const double* ym2){
        double dr_k = yn[0] - y[0];
        double dr_g = y[0] - P.ground;
        double prev[E];
        #pragma unroll
        for(int e = 0; e < E; e++) prev[e] = y[e] + (y[e] - yn[e]) / dr_k * dr_g;
        Medium mr = medium_at(P, prev[0]);
        double c_ref = mr.c;
        double dnu_r_ds = -1.0 / c_ref * (C.c0 / c_ref * mr.dc + prev[4] * mr.dv + prev[5] * mr.du
                                          + c_ref / prev[0] * (prev[4] * prev[4] + prev[5] * prev[5]));
        #pragma unroll
        for(int e = 0; e < E; e++) y[e] = prev[e];
        y[0] = P.ground;
        y[3] = -prev[3];
        if(AMP){
            y[6] = -prev[6]; y[12] = -prev[12];
            double den = c_ref / C.c0 * prev[3];
            y[9]  = -prev[9]  + 2.0 * dnu_r_ds * prev[6]  / den;
            y[15] = -prev[15] + 2.0 * dnu_r_ds * prev[12] / den;
        }
    }
    // one path segment: travel time (Global.cpp:527-589) and attenuation (Global.cpp:634-670, sin(lat) in ds: Q3); A, B = its two path rows
    static DEVINL void pp_aux(const GeoacDevParams& P, const double* st, size_t np, double* aux){}
    // k_postpass_tab: the segment's geometry (G.x = the abscissa of its midpoint) and, once the medium there is known, its travel time
    static DEVINL void pp_geom(const GeoacDevParams& P, const double* aux, const double* A, const double* B, PPGeom& G, double* ref = nullptr){
        double ar = A[0], at_ = A[1], ap = A[2], an0 = A[3], an1 = A[4], an2 = A[5];
        double dr = B[0] - ar, dt = B[1] - at_, dp = B[2] - ap;
        double r = ar + dr / 2.0, t = at_ + dt / 2.0;
        // sin / cos of the midpoint latitude.  ref (k_postpass_tab; angle, sin, cos): from the multiple of 2^-7 rad next to it - a function of the
        // segment alone, whatever thread holds it - by a rotation of at most 2^-8 rad (rot_fifth: exact to 1e-19); a thread's consecutive segments
        // share that point for ~50 km of travel, so the full routine (70 instructions) runs once per thread and crossing
        double sn, cs;
        if(ref){
            const double tq = __builtin_rint(t * 128.0) * (1.0 / 128.0);
            if(__builtin_expect(tq != ref[0], 0)){
                // (a table of the device's own fsincos at these points, k_atab_build: inlined here the routine's fifteen constants were hoisted
                //  out of the segment loop and cost k_postpass_tab 20 registers for a block that runs once per ~50 km of travel)
                ref[0] = tq;
                int iq = (int)(tq * 128.0) + GEOAC_LAT_OFF;          // (ref is only passed with the table on: lat_trig is there)
                iq = iq < 0 ? 0 : (iq > GEOAC_LAT_N - 1 ? GEOAC_LAT_N - 1 : iq);
                ref[1] = P.lat_trig[2 * iq]; ref[2] = P.lat_trig[2 * iq + 1];
            }
            rot_fifth(ref[1], ref[2], t - tq, sn, cs);
        } else fsincos(t, sn, cs);
        double rdt = r * dt;
        double e1 = r * cs * dp, e2 = r * sn * dp;
        G.ds_tt = fsqrt(dr * dr + rdt * rdt + e1 * e1);
        G.ds_at = fsqrt(dr * dr + rdt * rdt + e2 * e2);
        G.a0 = an0 + (B[3] - an0) / 2.0; G.a1 = an1 + (B[4] - an1) / 2.0; G.a2 = an2 + (B[5] - an2) / 2.0;
        G.a3 = frsq(G.a0 * G.a0 + G.a1 * G.a1 + G.a2 * G.a2);       // 1 / |nu|
        G.x = r;
    }
    static DEVINL double pp_tt(const GeoacDevParams& P, const double* aux, const PPGeom& G, double T, double u, double v){
        double qT = kGamR * T;
        double cn = (qT * frsq(qT)) * G.a3;
        double cp0 = cn * G.a0, cp1 = cn * G.a1 + v, cp2 = cn * G.a2 + u;
        return G.ds_tt * frsq(cp0 * cp0 + cp1 * cp1 + cp2 * cp2);
    }
    // the exact post-pass (k_postpass; GEOAC_ABS_TABLE=0): SuthBass_Alpha evaluated at the midpoint
    static DEVINL void segment(const GeoacDevParams& P, const double* st, size_t np, const double* a, const double* b, double& tt, double& at){
        double A[6], B[6];
        #pragma unroll
        for(int c = 0; c < 6; c++){ A[c] = a[c * np]; B[c] = b[c * np]; }
        pp_exact<EqGlobal<AMP_>>(P, nullptr, A, B, tt, at);
    }
};


// Two lanes per ray for the Global set with amplitudes: lanes 2j and 2j+1 both carry the base ray j (identical
// instructions on identical data, so every decision is taken identically) and ONE of the two launch-angle derivative
// systems each (lane parity 0: d/d(inclination), 1: d/d(azimuth)) - the two systems are the same code on different
// data (Global.cpp:273-367 is written twice, once per angle).  Halves the auxiliary work on the serial chain and puts
// a wave on every SIMD for the 32 400-ray metric fan.  Lanes exchange values only at leg ends (arrival record).
struct EqGlobalPair : EqGlobal<true> {
    using Full = EqGlobal<true>;
    static constexpr int E = 12, LANES = 2;
    static constexpr bool SPLIT = true; static constexpr bool ROW_SPLIT = true;      // (the two lanes also store half a path row each)
    template <typename TabPtr>
    static DEVINL void rhs(TabPtr tab, const GeoacDevParams& P, int& seg, const RayCtx& C, const double* y0, const double* yt, int stage, double* dy){
        if(stage == 0) global_rhs<true, 1, true>(tab, P, seg, C.rec, yt, C.cur[0], C.cur[1], 0.0, dy, C.rcp0);
        else global_rhs<true, 1>(tab, P, seg, C.rec, yt, C.cur[0], C.cur[1], yt[1] - y0[1], dy, C.rcp0);
    }
    // reflection of the base ray and of this lane's derivative system (Global.cpp:140-205, Q1 linear intercept)
    static DEVINL void reflect(const GeoacDevParams& P, const RayCtx& C, const double* yn, double* y, const double* ym2){
        double dr_k = yn[0] - y[0];
        double dr_g = y[0] - P.ground;
        double prev[E];
        #pragma unroll
        for(int e = 0; e < E; e++) prev[e] = y[e] + (y[e] - yn[e]) / dr_k * dr_g;
        Medium mr = medium_at(P, prev[0]);
        double c_ref = mr.c;
        double dnu_r_ds = -1.0 / c_ref * (C.c0 / c_ref * mr.dc + prev[4] * mr.dv + prev[5] * mr.du
                                          + c_ref / prev[0] * (prev[4] * prev[4] + prev[5] * prev[5]));
        #pragma unroll
        for(int e = 0; e < E; e++) y[e] = prev[e];
        y[0] = P.ground;
        y[3] = -prev[3];
        y[6] = -prev[6];
        y[9] = -prev[9] + 2.0 * dnu_r_ds * prev[6] / (c_ref / C.c0 * prev[3]);
    }
};

#pragma clang fp contract(fast)          // grid sets: as before (their kernels are compared with each other to rounding, see DESIGN)
#include "geoac_rngdep.h"
#ifndef GEOAC_CELL_REGS
#define GEOAC_CELL_REGS(C) ((CACHE_ || !COOP_) ? (C).cell : nullptr)   // node coordinates of the hinted cell in registers (not in the cooperative kernels: no registers to spare)
#endif
#ifndef GEOAC_PP_TILE
#define GEOAC_PP_TILE 1               // grid sets: post-pass over 16-row x 16-ray tiles (0: one row x 256 rays per workgroup, A/B builds)
#endif
#ifndef GEOAC_PP_DEDUP
#define GEOAC_PP_DEDUP 1              // Cartesian grid set: the post-pass reads the table once per distinct key of a wave (k_postpass)
#endif
#define GEOAC_PP_SLOTS 6              // keys per round
#define GEOAC_PP_SLOTB 2064           // 16 half-records x 128 B + 16 B pad (broadcast reads of different slots fall into different banks)

// Range-dependent Cartesian set (GeoAc3D.RngDep): EquationSets.3DRngDep.cpp + G2S_MultiDimSpline3D.cpp
template <bool AMP_, int NL_ = 1, bool COOP_ = false, bool CACHE_ = false> struct Eq3DRngDep {
    static constexpr bool SEG1D = false;
    static constexpr int AUX_SAVE = 0;
    static constexpr bool CACHE = CACHE_;                           // NL_ = 4, small fans: per-lane record cache and z nodes in LDS (grid_cache_fill)
    static constexpr bool AMP = AMP_;
    static constexpr bool COOP = COOP_;                             // wave-cooperative record gather through LDS (grid_eval3_glds / _coop8): NL_ = 1, every lane of the wave stays in the loop
    static constexpr int XCHG_BYTES = (GRec<false>::PACKED && GEOAC_COOP_GLDS) ? GEOAC_GLDS_BYTES : 64 * GEOAC_COOP_SLOT;   // per wave: LDS-DMA ring / exchange slots
    static constexpr bool PP_TILE = GEOAC_PP_TILE;                  // post-pass over 16-row x 16-ray tiles (k_postpass)
    static constexpr bool PP_DEDUP = GEOAC_PP_TILE && GEOAC_PP_DEDUP;  // and one table read per distinct (cell, segment) key of a wave
    static constexpr bool PP_TWO = false, GLOBAL = false;
    static constexpr int SYS_SHIFT = 0; static constexpr bool ROW_SPLIT = false;
    static constexpr bool LDS_STATE = !(CACHE_ && GEOAC_CACHE_REG_STATE);   // the step's rows y and yn live in LDS while the four stages run (GEOAC_CACHE_REG_STATE: the record-cache kernels could keep them in registers since the arrival evaluation left k_rk4 - measured slower on the ring)
    static constexpr int NB = 6, NS = 6;
    static constexpr int PP_WAVES = 3;
    static constexpr int E = AMP_ ? 18 : 6, PW = 6, HIDX = 2, LANES = NL_;            // NL_ = 1, 2 or 4 lanes per ray, each evaluating 4/NL_ of the cell corners (identical state otherwise)
    static constexpr bool SPLIT = false;
    static constexpr bool KM2 = true, HMAX_PER_LEG = true;          // quadratic intercept; turning height per leg (Q8)
    static constexpr bool KM2_MEM = true;                           // row k-2 lives in the state block, not in 36 registers: this kernel spills as it is, and the row is only read at a reflection

    // SuthBass reference state at (0, 0, z_grnd) (Atmo_State.Absorption.cpp:31-33), once per fan
    static DEVINL void fan_init(const GeoacDevParams& P){
        Medium3 g = medium3_at<true, false>(P, 0.0, 0.0, P.z_grnd);
        double cm = g.c * 1000.0;
        P.dev_consts[0] = cm * cm / (kRgas * kGam);
        P.dev_consts[1] = g.rho * (cm * cm) / kGam * 1000.0;
        P.dev_consts[2] = cbrt(P.dev_consts[0]);
    }
    // GeoAc_SetInitialConditions: 3DRngDep.cpp:70-136
    static DEVINL void init(const GeoacDevParams& P, double th, double ph, double* y, RayCtx& C){
        double z0 = P.z_grnd < P.src[2] ? P.src[2] : P.z_grnd;       // GeoAc3D.RngDep_main.cpp:165
        Medium3 m = medium3_at<false, false>(P, P.src[0], P.src[1], z0);
        double c0 = m.c;
        double Mc[3] = { m.u / c0, m.v / c0, 0.0 };
        double sth = sin(th), cth = cos(th), sph = sin(ph), cph = cos(ph);
        double nu0[3] = { cth * cph, cth * sph, sth };
        double mth[3] = { -sth * cph, -sth * sph, cth };
        double mph[3] = { -cth * sph, cth * cph, 0.0 };
        double MS = 1.0 + (nu0[0] * Mc[0] + nu0[1] * Mc[1] + nu0[2] * Mc[2]);
        C.c0 = c0; C.nu0 = 1.0 / MS;
        for(int q = 0; q < 6; q++) C.a[q] = 0.0;
        y[0] = P.src[0]; y[1] = P.src[1]; y[2] = z0;
        for(int e = 0; e < 3; e++) y[3 + e] = nu0[e] / MS;
        if(AMP){
            double dt = mth[0] * Mc[0] + mth[1] * Mc[1] + mth[2] * Mc[2];
            double dp = mph[0] * Mc[0] + mph[1] * Mc[1] + mph[2] * Mc[2];
            for(int e = 0; e < 3; e++){
                y[9 + e]  = mth[e] / MS - nu0[e] / (MS * MS) * dt;
                y[15 + e] = mph[e] / MS - nu0[e] / (MS * MS) * dp;
            }
        }
    }
    static DEVINL double height(const GeoacDevParams& P, const double* y){ return y[2]; }
    static DEVINL double above_ground(const GeoacDevParams& P, const double* y){ return y[2] - P.ground; }
    template <typename TabPtr>
    static DEVINL void rhs(TabPtr tab, const GeoacDevParams& P, int& seg, const RayCtx& C, const double* y0, const double* yt, int stage, double* dy){
        rngdep_rhs<AMP, NL_, COOP_, CACHE_>(P, seg, yt, dy, (int)(threadIdx.x & (NL_ - 1)), (char*)tab, &C.ckey, &C.kxy, GEOAC_CELL_REGS(C));
    }
    // 3DRngDep.cpp:451-472
    static DEVINL void checks(const GeoacDevParams& P, RayCtx& C, const double* y, const double* yn, int k, bool& brk, bool& gnd){
        brk = (yn[0] > P.xy_lim[1]) || (yn[0] < P.xy_lim[0]) || (yn[1] > P.xy_lim[3]) || (yn[1] < P.xy_lim[2]) || (yn[2] > P.vert_limit);
        gnd = yn[2] < P.ground;
    }
    static DEVINL void accept(RayCtx& C){}
    static DEVINL void resume(const GeoacDevParams& P, RayCtx& C, const double* y){}
    static DEVINL void restart(const GeoacDevParams& P, RayCtx& C, const double* y){}
    // GeoAc3D.RngDep_main.cpp:298-301
    static DEVINL void arrival(const GeoacDevParams& P, const RayCtx& C, int slot, const double* yn, double* R){
        Medium3 mg = medium3_at<false, false>(P, yn[0], yn[1], P.z_grnd);
        double incl = -asin(mg.c / C.c0 * yn[5]) * 180.0 / kPi;
        double baz = 90.0 - atan2(-yn[4], -yn[3]) * 180.0 / kPi;
        while(baz < -180.0) baz += 360.0;
        while(baz > 180.0) baz -= 360.0;
        R[GEOAC_REC_INCL] = incl;
        R[GEOAC_REC_BACKAZ] = baz;
        R[GEOAC_REC_RANGE] = sqrt(yn[0] * yn[0] + yn[1] * yn[1]);
        if(AMP){
            double amp, D;
            amp_jac(P, C, slot, yn, amp, D);
            R[GEOAC_REC_AMP] = amp;
            R[GEOAC_REC_JACOB] = D;
        }
    }
    // GeoAc_Jacobian / GeoAc_Amplitude: 3DRngDep.cpp:547-592
    static DEVINL void amp_jac(const GeoacDevParams& P, const RayCtx& C, int slot, const double* yn, double& amp, double& D){
        double z0 = P.z_grnd < P.src[2] ? P.src[2] : P.z_grnd;
        Medium3 m = medium3_at<true, false>(P, yn[0], yn[1], yn[2]);
        Medium3 m0 = medium3_at<true, false>(P, P.src[0], P.src[1], z0);
        double n0 = yn[3], n1 = yn[4], n2 = yn[5];
        double nmag = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
        double cp0 = m.c * n0 / nmag + m.u, cp1 = m.c * n1 / nmag + m.v, cp2 = m.c * n2 / nmag;
        double cpm = sqrt(cp0 * cp0 + cp1 * cp1 + cp2 * cp2);
        double dxds = cp0 / cpm, dyds = cp1 / cpm, dzds = cp2 / cpm;
        D = dxds * (yn[7] * yn[14] - yn[13] * yn[8]) - yn[6] * (dyds * yn[14] - dzds * yn[13]) + yn[12] * (dyds * yn[8] - dzds * yn[7]);
        double th_l = P.theta_deg[slot] * kPi / 180.0, ph_l = kPi / 2.0 - P.phi_deg[slot] * kPi / 180.0;
        double c0 = C.c0;
        double nu_mag = (c0 - n0 * m.u - n1 * m.v) / m.c;
        double nu_mag0 = 1.0 - n0 * m0.u / c0 - n1 * m0.v / c0;
        double ap0 = m.c * n0 / nu_mag + m.u, ap1 = m.c * n1 / nu_mag + m.v, ap2 = m.c * n2 / nu_mag;
        double aq0 = c0 * cos(th_l) * cos(ph_l) + m0.u, aq1 = c0 * cos(th_l) * sin(ph_l) + m0.v, aq2 = c0 * sin(th_l);
        double apm = sqrt(ap0 * ap0 + ap1 * ap1 + ap2 * ap2);
        double aqm = sqrt(aq0 * aq0 + aq1 * aq1 + aq2 * aq2);
        double num = m.rho * nu_mag * (m.c * m.c * m.c) * aqm * cos(th_l);
        double den = m0.rho * nu_mag0 * (c0 * c0 * c0) * apm * D;
        amp = 1.0 / (4.0 * kPi) * sqrt(fabs(num / den));
    }
    // ApproximateIntercept + SetReflectionConditions: 3DRngDep.cpp:142-201
    static DEVINL void reflect(const GeoacDevParams& P, const RayCtx& C, const double* yn, double* y, const double* ym2){
        double dz_k = yn[2] - y[2];
        double dz_g = y[2] - P.ground;
        double prev[E];
        #pragma unroll
        for(int e = 0; e < E; e++)
            prev[e] = y[e] + (y[e] - yn[e]) / dz_k * dz_g + 1.0 / 2.0 * (yn[e] + ym2[e] - 2.0 * y[e]) / (dz_k * dz_k) * (dz_g * dz_g);
        Medium3 mg = medium3_at<false, true>(P, prev[0], prev[1], P.z_grnd);
        double dnuz_ds = -1.0 / mg.c * (C.c0 / mg.c * mg.dcz + prev[3] * mg.duz + prev[4] * mg.dvz);
        #pragma unroll
        for(int e = 0; e < E; e++) y[e] = prev[e];
        y[2] = P.ground;
        y[5] = -prev[5];
        if(AMP){
            y[8] = -prev[8]; y[14] = -prev[14];
            double den = mg.c / C.c0 * prev[5];
            y[11] = -prev[11] + 2.0 * dnuz_ds * prev[8]  / den;
            y[17] = -prev[17] + 2.0 * dnuz_ds * prev[14] / den;
        }
    }
    // 3DRngDep.cpp:478-542, 598-634
    // one path segment in two halves (k_postpass fetches the medium of the midpoint itself, one table read per distinct key of a wave)
    struct SegGeom { double x, y, z, ds, n0, n1, n2, inm; };
    static DEVINL void seg_mid(const GeoacDevParams& P, size_t np, const double* a, const double* b, SegGeom& G){
        double ax = a[0], ay = a[np], az = a[2 * np], an0 = a[3 * np], an1 = a[4 * np], an2 = a[5 * np];
        double dx = b[0] - ax, dy = b[np] - ay, dz = b[2 * np] - az;
        G.ds = fsqrt(dx * dx + dy * dy + dz * dz);
        G.x = ax + dx / 2.0; G.y = ay + dy / 2.0; G.z = az + dz / 2.0;
        G.n0 = an0 + (b[3 * np] - an0) / 2.0; G.n1 = an1 + (b[4 * np] - an1) / 2.0; G.n2 = an2 + (b[5 * np] - an2) / 2.0;
        G.inm = frsq(G.n0 * G.n0 + G.n1 * G.n1 + G.n2 * G.n2);
    }
    static DEVINL void seg_sums(const GeoacDevParams& P, const SegGeom& G, const Medium3& m, const Medium3& /*g: spherical set only*/, double& tt, double& at){
        double cn = m.c * G.inm;
        double cp0 = cn * G.n0 + m.u, cp1 = cn * G.n1 + m.v, cp2 = cn * G.n2;
        tt = G.ds * frsq(cp0 * cp0 + cp1 * cp1 + cp2 * cp2);
        at = suthbass_alpha(P, G.z, m.c, m.rho, P.freq, P.dev_consts[0], P.dev_consts[1], P.dev_consts[2]) * G.ds;
    }
    static DEVINL void segment(const GeoacDevParams& P, const double* st, size_t np, const double* a, const double* b, double& tt, double& at){
        SegGeom G; seg_mid(P, np, a, b, G);
        Medium3 m = medium3_at<true, false>(P, G.x, G.y, G.z);
        seg_sums(P, G, m, m, tt, at);
    }
};

// Eight lanes per ray for the Cartesian grid set with amplitudes, small fans (eigenray rounds): cell corner q & 3, launch-angle system q >> 2
// (see EqGlobalRngDepOct).  Row k - 2 of the quadratic intercept stays in the state block (KM2_MEM), each system's part stored by its first lane.
struct Eq3DRngDepOct : Eq3DRngDep<true, 4, false, true> {
    using Full = Eq3DRngDep<true, 4, false, true>;
    static constexpr int E = 12, LANES = 8;
    static constexpr bool LDS_STATE = GEOAC_OCT_LDS_STATE != 0;
    static constexpr bool SPLIT = true; static constexpr bool ROW_SPLIT = false;
    static constexpr int SYS_SHIFT = 2;
    template <typename TabPtr>
    static DEVINL void rhs(TabPtr tab, const GeoacDevParams& P, int& seg, const RayCtx& C, const double* y0, const double* yt, int stage, double* dy){
        rngdep_rhs<true, 4, false, true, 1>(P, seg, yt, dy, (int)(threadIdx.x & 3), (char*)tab, &C.ckey, &C.kxy, C.cell);
    }
    // quadratic intercept + reflection of the base ray and of this lane's system: 3DRngDep.cpp:146-216
    static DEVINL void reflect(const GeoacDevParams& P, const RayCtx& C, const double* yn, double* y, const double* ym2){
        double dz_k = yn[2] - y[2];
        double dz_g = y[2] - P.ground;
        double prev[E];
        #pragma unroll
        for(int e = 0; e < E; e++)
            prev[e] = y[e] + (y[e] - yn[e]) / dz_k * dz_g + 1.0 / 2.0 * (yn[e] + ym2[e] - 2.0 * y[e]) / (dz_k * dz_k) * (dz_g * dz_g);
        Medium3 mg = medium3_at<false, true>(P, prev[0], prev[1], P.z_grnd);
        double dnuz_ds = -1.0 / mg.c * (C.c0 / mg.c * mg.dcz + prev[3] * mg.duz + prev[4] * mg.dvz);
        #pragma unroll
        for(int e = 0; e < E; e++) y[e] = prev[e];
        y[2] = P.ground;
        y[5] = -prev[5];
        y[8] = -prev[8];
        double den = mg.c / C.c0 * prev[5];
        y[11] = -prev[11] + 2.0 * dnuz_ds * prev[8] / den;
    }
};

// Range-dependent spherical set (GeoAcGlobal.RngDep): EquationSets.GlobalRngDep.cpp + G2S_GlobalMultiDimSpline3D.cpp.
// Grid axes in table order: x = latitude, y = longitude [rad], z = geocentric radius; xy_lim = lat/lon box of the break check.
template <bool AMP_, int NL_ = 1, bool COOP_ = false, bool CACHE_ = false> struct EqGlobalRngDep {
    static constexpr bool SEG1D = false;
    static constexpr int AUX_SAVE = 2;                              // sin / cos of the latitude, carried along the ray
    static constexpr bool CACHE = CACHE_;
    static constexpr bool AMP = AMP_;
    static constexpr bool COOP = COOP_;
    static constexpr int XCHG_BYTES = GEOAC_COOP_GLDS ? GEOAC_GLDS_BYTES : 64 * GEOAC_COOP_SLOT;
    static constexpr bool PP_TILE = GEOAC_PP_TILE;
    static constexpr bool PP_DEDUP = GEOAC_PP_TILE && GEOAC_PP_DEDUP;
    static constexpr bool PP_TWO = true, GLOBAL = true;             // (a second point per segment: the absorption's reference state at ground level)
    static constexpr int SYS_SHIFT = 0; static constexpr bool ROW_SPLIT = false;
    static constexpr bool LDS_STATE = !(CACHE_ && GEOAC_CACHE_REG_STATE);
    static constexpr int NB = 6, NS = 6;
    static constexpr int PP_WAVES = 2;                                 // post-pass at two waves per SIMD: 0 scratch (168 registers spill 108 B: the absorption integrand AND its ground-level reference state)
    static constexpr int E = AMP_ ? 18 : 6, PW = 6, HIDX = 0, LANES = NL_;
    static constexpr bool SPLIT = false;
    static constexpr bool KM2 = false, KM2_MEM = false, HMAX_PER_LEG = true;         // linear intercept (Q1, GlobalRngDep.cpp:147-148); turning height per leg (Q8)

    static DEVINL Medium as_medium(const Medium3& g){ Medium m; m.c = g.c; m.dc = g.dcz; m.u = g.u; m.du = g.duz; m.v = g.v; m.dv = g.dvz; m.rho = g.rho; return m; }
    static DEVINL void fan_init(const GeoacDevParams& P){}
    // GeoAc_SetInitialConditions: GlobalRngDep.cpp:77-137
    static DEVINL void init(const GeoacDevParams& P, double th, double ph, double* y, RayCtx& C){
        double z_src = P.src[0] < P.z_grnd ? P.z_grnd : P.src[0];     // GeoAcGlobal.RngDep_main.cpp:177
        double r0 = z_src + P.r_earth;
        double lat0 = P.src[1] * kPi / 180.0, lon0 = P.src[2] * kPi / 180.0;
        Medium3 m = medium3_at<false, false, true>(P, lat0, lon0, r0);
        double c0 = m.c;
        double Mach[3] = { 0.0, m.v / c0, m.u / c0 };
        double sth = sin(th), cth = cos(th), sph = sin(ph), cph = cos(ph);
        double nu0[3]  = { sth, cth * sph, cth * cph };
        double mlt[3]  = { cth, -sth * sph, -sth * cph };
        double mlp[3]  = { 0.0, cth * cph, -cth * sph };
        double MS = 1.0 + (nu0[0] * Mach[0] + nu0[1] * Mach[1] + nu0[2] * Mach[2]);
        y[0] = r0; y[1] = lat0; y[2] = lon0;
        for(int e = 0; e < 3; e++) y[3 + e] = nu0[e] / MS;
        if(AMP){
            double dlt = mlt[0] * Mach[0] + mlt[1] * Mach[1] + mlt[2] * Mach[2];
            double dlp = mlp[0] * Mach[0] + mlp[1] * Mach[1] + mlp[2] * Mach[2];
            for(int e = 0; e < 3; e++){
                y[9 + e]  = mlt[e] / MS - nu0[e] / (MS * MS) * dlt;
                y[15 + e] = mlp[e] / MS - nu0[e] / (MS * MS) * dlp;
            }
        }
        C.c0 = c0; C.nu0 = 1.0 / MS;
        fsincos(lat0, C.a[0], C.a[1]);
        C.a[2] = 0.0; C.a[3] = 1.0; C.a[4] = 0.0; C.a[5] = 0.0;
    }
    static DEVINL double height(const GeoacDevParams& P, const double* y){ return y[0] - P.r_earth; }
    static DEVINL double above_ground(const GeoacDevParams& P, const double* y){ return y[0] - P.ground; }
    template <typename TabPtr>
    static DEVINL void rhs(TabPtr tab, const GeoacDevParams& P, int& seg, const RayCtx& C, const double* y0, const double* yt, int stage, double* dy){
        double s2, c2;
        rot_small(C.a[0], C.a[1], yt[1] - y0[1], s2, c2);          // sin/cos(lat) carried along the ray, as in EqGlobal
        globalrd_rhs<AMP, NL_, COOP_, CACHE_>(P, seg, yt, s2, c2, dy, (int)(threadIdx.x & (NL_ - 1)), (char*)tab, &C.ckey, &C.kxy, GEOAC_CELL_REGS(C));
    }
    // GeoAc_BreakCheck / GeoAc_GroundCheck: GlobalRngDep.cpp:523-545
    static DEVINL void checks(const GeoacDevParams& P, RayCtx& C, const double* y, const double* yn, int k, bool& brk, bool& gnd){
        if((k & 63) == 0) fsincos(yn[1], C.t[0], C.t[1]);          // periodic exact re-sync
        else rot_small(C.a[0], C.a[1], yn[1] - y[1], C.t[0], C.t[1]);
        brk = (yn[0] > P.vert_limit) || (yn[1] < P.xy_lim[0]) || (yn[1] > P.xy_lim[1]) || (yn[2] < P.xy_lim[2]) || (yn[2] > P.xy_lim[3]);
        gnd = yn[0] < P.ground;
    }
    static DEVINL void accept(RayCtx& C){ C.a[0] = C.t[0]; C.a[1] = C.t[1]; }
    static DEVINL void resume(const GeoacDevParams& P, RayCtx& C, const double* y){}
    static DEVINL void restart(const GeoacDevParams& P, RayCtx& C, const double* y){ fsincos(y[1], C.a[0], C.a[1]); }
    // arrival row: GeoAcGlobal.RngDep_main.cpp:304-329 (inclination without the leading minus: Q10)
    static DEVINL void arrival(const GeoacDevParams& P, const RayCtx& C, int slot, const double* yn, double* R){
        const double lat0 = P.src[1] * kPi / 180.0, lon0 = P.src[2] * kPi / 180.0;
        Medium3 m = medium3_at<false, false, true>(P, yn[1], yn[2], yn[0]);
        double incl = asin(m.c / C.c0 * yn[3]) * 180.0 / kPi;
        double baz = 90.0 - atan2(-yn[4], -yn[5]) * 180.0 / kPi;
        if(baz < -180.0) baz += 360.0;
        if(baz > 180.0) baz -= 360.0;
        double g1 = sin((yn[1] - lat0) / 2.0); g1 *= g1;
        double g2 = sin((yn[2] - lon0) / 2.0); g2 = cos(lat0) * cos(yn[1]) * g2 * g2;
        R[GEOAC_REC_INCL] = incl;
        R[GEOAC_REC_BACKAZ] = baz;
        R[GEOAC_REC_RANGE] = 2.0 * P.r_earth * asin(sqrt(g1 + g2));
        if(AMP){
            double amp, D;
            amp_jac(P, C, slot, yn, amp, D);
            R[GEOAC_REC_AMP] = amp;
            R[GEOAC_REC_JACOB] = D;
        }
    }
    // GeoAc_Jacobian / GeoAc_Amplitude: GlobalRngDep.cpp:617-652 (same expressions as the stratified set, medium from the grid)
    static DEVINL void amp_jac(const GeoacDevParams& P, const RayCtx& C, int slot, const double* yr, double& amp, double& D){
        double z_src = P.src[0] < P.z_grnd ? P.z_grnd : P.src[0];
        const double lat0 = P.src[1] * kPi / 180.0, lon0 = P.src[2] * kPi / 180.0;
        Medium m  = as_medium(medium3_at<true, false, true>(P, yr[1], yr[2], yr[0]));
        Medium m0 = as_medium(medium3_at<true, false, true>(P, lat0, lon0, z_src + P.r_earth));
        D = global_jacobian(m, yr);
        double th_l = P.theta_deg[slot] * kPi / 180.0, ph_l = kPi / 2.0 - P.phi_deg[slot] * kPi / 180.0;
        amp = global_amplitude(P, m, m0, yr, C.c0, C.nu0, th_l, ph_l, D);
    }
    // ApproximateIntercept + SetReflectionConditions: GlobalRngDep.cpp:141-210
    static DEVINL void reflect(const GeoacDevParams& P, const RayCtx& C, const double* yn, double* y, const double* ym2){
        double dr_k = yn[0] - y[0];
        double dr_g = y[0] - P.ground;
        double prev[E];
        #pragma unroll
        for(int e = 0; e < E; e++) prev[e] = y[e] + (y[e] - yn[e]) / dr_k * dr_g;
        Medium3 mr = medium3_at<false, true, true>(P, prev[1], prev[2], prev[0]);
        double c_ref = mr.c;
        double dnu_r_ds = -1.0 / c_ref * (C.c0 / c_ref * mr.dcz + prev[4] * mr.dvz + prev[5] * mr.duz
                                          + c_ref / prev[0] * (prev[4] * prev[4] + prev[5] * prev[5]));
        #pragma unroll
        for(int e = 0; e < E; e++) y[e] = prev[e];
        y[0] = P.ground;
        y[3] = -prev[3];
        if(AMP){
            y[6] = -prev[6]; y[12] = -prev[12];
            double den = c_ref / C.c0 * prev[3];
            y[9]  = -prev[9]  + 2.0 * dnu_r_ds * prev[6]  / den;
            y[15] = -prev[15] + 2.0 * dnu_r_ds * prev[12] / den;
        }
    }
    // one path segment: GlobalRngDep.cpp:549-612 (travel time), 657-693 (attenuation, sin(lat) in ds: Q3).  SuthBass reference
    // state at radius z_grnd (clamps to the lowest node) and the LOCAL lat/lon (Atmo_State.Absorption.Global.cpp:31-32)
    // in two halves (k_postpass fetches the medium at the midpoint and at ground level itself, one table read per distinct key of a wave);
    // SegGeom in table order: x = latitude, y = longitude, z = radius of the midpoint
    struct SegGeom { double x, y, z, ds_tt, ds_at, n0, n1, n2, inm; };
    static DEVINL void seg_mid(const GeoacDevParams& P, size_t np, const double* a, const double* b, SegGeom& G){
        double ar = a[0], at_ = a[np], ap = a[2 * np], an0 = a[3 * np], an1 = a[4 * np], an2 = a[5 * np];
        double dr = b[0] - ar, dt = b[np] - at_, dp = b[2 * np] - ap;
        double r = ar + dr / 2.0, t = at_ + dt / 2.0, p = ap + dp / 2.0;
        double sn, cs; fsincos(t, sn, cs);
        double rdt = r * dt;
        double e1 = r * cs * dp, e2 = r * sn * dp;
        G.ds_tt = fsqrt(dr * dr + rdt * rdt + e1 * e1);
        G.ds_at = fsqrt(dr * dr + rdt * rdt + e2 * e2);
        G.n0 = an0 + (b[3 * np] - an0) / 2.0; G.n1 = an1 + (b[4 * np] - an1) / 2.0; G.n2 = an2 + (b[5 * np] - an2) / 2.0;
        G.inm = frsq(G.n0 * G.n0 + G.n1 * G.n1 + G.n2 * G.n2);
        G.x = t; G.y = p; G.z = r;
    }
    static DEVINL void seg_sums(const GeoacDevParams& P, const SegGeom& G, const Medium3& m, const Medium3& g, double& tt, double& at){
        double cn = m.c * G.inm;
        double cp0 = cn * G.n0, cp1 = cn * G.n1 + m.v, cp2 = cn * G.n2 + m.u;
        tt = G.ds_tt * frsq(cp0 * cp0 + cp1 * cp1 + cp2 * cp2);
        double cm = g.c * 1000.0;
        double T_o = cm * cm / (kRgas * kGam);
        double P_o = g.rho * (cm * cm) / kGam * 1000.0;
        at = suthbass_alpha(P, G.z - P.r_earth, m.c, m.rho, P.freq, T_o, P_o, cbrt(T_o)) * G.ds_at;
    }
    static DEVINL void segment(const GeoacDevParams& P, const double* st, size_t np, const double* a, const double* b, double& tt, double& at){
        SegGeom G; seg_mid(P, np, a, b, G);
        Medium3 m = medium3_at<true, false, true>(P, G.x, G.y, G.z);
        Medium3 g = medium3_at<true, false, true, false>(P, G.x, G.y, P.z_grnd);      // reference state: c and rho only
        seg_sums(P, G, m, g, tt, at);
    }
};


// Eight lanes per ray for the spherical grid set with amplitudes, small fans (the eigenray rounds): lane q of a ray evaluates cell corner
// q & 3 (as in the four-lane kernels: partial sums added across the quad) and carries the base ray plus ONE of the two launch-angle
// systems, q >> 2 (as EqGlobalPair does for the stratified set).  The stage of the four-lane kernel is 1 245 instructions, issue-bound,
// 270 of them the second system's right-hand side and state update; a fan of a few rays has lanes to spare.
struct EqGlobalRngDepOct : EqGlobalRngDep<true, 4, false, true> {
    using Full = EqGlobalRngDep<true, 4, false, true>;
    static constexpr int E = 12, LANES = 8;
    static constexpr bool LDS_STATE = GEOAC_OCT_LDS_STATE != 0;
    static constexpr bool SPLIT = true; static constexpr bool ROW_SPLIT = false;
    static constexpr int SYS_SHIFT = 2;
    template <typename TabPtr>
    static DEVINL void rhs(TabPtr tab, const GeoacDevParams& P, int& seg, const RayCtx& C, const double* y0, const double* yt, int stage, double* dy){
        double s2, c2;
        rot_small(C.a[0], C.a[1], yt[1] - y0[1], s2, c2);
        globalrd_rhs<true, 4, false, true, 1>(P, seg, yt, s2, c2, dy, (int)(threadIdx.x & 3), (char*)tab, &C.ckey, &C.kxy, C.cell);
    }
    // ApproximateIntercept + SetReflectionConditions of the base ray and of this lane's system: GlobalRngDep.cpp:141-210
    static DEVINL void reflect(const GeoacDevParams& P, const RayCtx& C, const double* yn, double* y, const double* ym2){
        double dr_k = yn[0] - y[0];
        double dr_g = y[0] - P.ground;
        double prev[E];
        #pragma unroll
        for(int e = 0; e < E; e++) prev[e] = y[e] + (y[e] - yn[e]) / dr_k * dr_g;
        Medium3 mr = medium3_at<false, true, true>(P, prev[1], prev[2], prev[0]);
        double c_ref = mr.c;
        double dnu_r_ds = -1.0 / c_ref * (C.c0 / c_ref * mr.dcz + prev[4] * mr.dvz + prev[5] * mr.duz
                                          + c_ref / prev[0] * (prev[4] * prev[4] + prev[5] * prev[5]));
        #pragma unroll
        for(int e = 0; e < E; e++) y[e] = prev[e];
        y[0] = P.ground;
        y[3] = -prev[3];
        y[6] = -prev[6];
        double den = c_ref / C.c0 * prev[3];
        y[9] = -prev[9] + 2.0 * dnu_r_ds * prev[6] / den;
    }
};

// Sixteen lanes per ray for the same fans when they are smaller still (at most four rays per wave and one wave per CU: the refinement rounds of an
// eigenray search are a handful of rays): lane q evaluates cell corner q & 3 of ONE field, (q >> 2) & 3, and carries the base ray plus the
// launch-angle system q >> 3 (globalrd_rhs<..., FSPLIT>).  A third of the table evaluation per lane; the fields change hands through LDS.
struct EqGlobalRngDepHex : EqGlobalRngDepOct {
    static constexpr int LANES = 16;
    static constexpr int SYS_SHIFT = 3;
#ifndef GEOAC_HEX_LDS_STATE
#define GEOAC_HEX_LDS_STATE 0
#endif
    static constexpr bool LDS_STATE = GEOAC_HEX_LDS_STATE != 0;    // the step's rows in registers: a third of the evaluation leaves room for them, and 24 LDS operations per stage go
    template <typename TabPtr>
    static DEVINL void rhs(TabPtr tab, const GeoacDevParams& P, int& seg, const RayCtx& C, const double* y0, const double* yt, int stage, double* dy){
        double s2, c2;
        rot_small(C.a[0], C.a[1], yt[1] - y0[1], s2, c2);
        globalrd_rhs<true, 4, false, true, 1, true>(P, seg, yt, s2, c2, dy, (int)(threadIdx.x & 3), (char*)tab, &C.ckey, &C.kxy, C.cell);
    }
};

// ... and the amplitude-less fans of that size - the inclination scans of an eigenray search once the receivers are few: the same split of the
// table evaluation over sixteen lanes (every lane carries the whole six-component ray; nothing to split there)
struct EqGlobalRngDepScan16 : EqGlobalRngDep<false, 4, false, true> {
    static constexpr int LANES = 16;
    template <typename TabPtr>
    static DEVINL void rhs(TabPtr tab, const GeoacDevParams& P, int& seg, const RayCtx& C, const double* y0, const double* yt, int stage, double* dy){
        double s2, c2;
        rot_small(C.a[0], C.a[1], yt[1] - y0[1], s2, c2);
        globalrd_rhs<false, 4, false, true, 2, true>(P, seg, yt, s2, c2, dy, (int)(threadIdx.x & 3), (char*)tab, &C.ckey, &C.kxy, C.cell);
    }
};

#pragma clang fp contract(off)           // stratified sets again
template <bool AMP_> struct Eq3D {
    static constexpr bool AMP = AMP_;
    static constexpr bool SEG1D = true;
    static constexpr int AUX_SAVE = 0;                              // (nu_x, nu_y and their launch-angle derivatives: constants of the ray)
    static constexpr bool COOP = false; static constexpr bool CACHE = false; static constexpr bool LDS_STATE = false; static constexpr int XCHG_BYTES = 0; static constexpr bool PP_TILE = false; static constexpr bool PP_DEDUP = false; static constexpr int SYS_SHIFT = 0; static constexpr bool ROW_SPLIT = false;
    static constexpr int PP_WAVES = 4;                              // post-pass at four waves per SIMD (127 registers, 52 B of spill): GeoAc3D 360 x 90 fan 185 -> 168 ms
    static constexpr int E = AMP_ ? 12 : 4, PW = 4, HIDX = 2, LANES = 1;
    static constexpr bool SPLIT = false;
    static constexpr int NB = 4, NS = 4;
    static constexpr bool KM2 = true, KM2_MEM = false, HMAX_PER_LEG = false;              // quadratic intercept needs row k-2

    // GeoAc_SetInitialConditions: EquationSets.3DStratified.cpp:69-131
    static DEVINL void init(const GeoacDevParams& P, double th, double ph, double* y, RayCtx& C){
        double z0 = P.z_grnd < P.src[2] ? P.src[2] : P.z_grnd;       // GeoAc3D_main.cpp:152
        Medium m = medium_at(P, z0);
        double c0 = m.c;
        double Mc[3] = { m.u / c0, m.v / c0, 0.0 };
        double sth = sin(th), cth = cos(th), sph = sin(ph), cph = cos(ph);
        double nu0[3] = { cth * cph, cth * sph, sth };
        double mth[3] = { -sth * cph, -sth * sph, cth };
        double mph[3] = { -cth * sph, cth * cph, 0.0 };
        double M = 1.0 + (nu0[0] * Mc[0] + nu0[1] * Mc[1] + nu0[2] * Mc[2]);
        double dMt = mth[0] * Mc[0] + mth[1] * Mc[1] + mth[2] * Mc[2];
        double dMp = mph[0] * Mc[0] + mph[1] * Mc[1] + mph[2] * Mc[2];
        C.c0 = c0; C.nu0 = 0.0;
        C.a[0] = nu0[0] / M; C.a[1] = nu0[1] / M;
        C.a[2] = mth[0] / M - nu0[0] / (M * M) * dMt;  C.a[3] = mth[1] / M - nu0[1] / (M * M) * dMt;
        C.a[4] = mph[0] / M - nu0[0] / (M * M) * dMp;  C.a[5] = mph[1] / M - nu0[1] / (M * M) * dMp;
        y[0] = P.src[0]; y[1] = P.src[1]; y[2] = z0; y[3] = nu0[2] / M;
        if(AMP){
            y[7]  = mth[2] / M - nu0[2] / (M * M) * dMt;
            y[11] = mph[2] / M - nu0[2] / (M * M) * dMp;
        }
    }
    static DEVINL void fan_init(const GeoacDevParams& P){}
    static DEVINL double height(const GeoacDevParams& P, const double* y){ return y[2]; }
    static DEVINL double above_ground(const GeoacDevParams& P, const double* y){ return y[2] - P.ground; }
    template <typename TabPtr>
    static DEVINL void rhs(TabPtr tab, const GeoacDevParams& P, int& seg, const RayCtx& C, const double* y0, const double* yt, int stage, double* dy){
        cart3_rhs<AMP>(tab, P, seg, C, yt, dy);
    }
    // 3DStratified.cpp:327-343
    static DEVINL void checks(const GeoacDevParams& P, RayCtx& C, const double* y, const double* yn, int k, bool& brk, bool& gnd){
        // r = sqrt(x^2 + y^2) > limit, decided on the square except within 1e-12 of the limit (there the reference's own expression): the root
        // is ~20 instructions on the serial chain of every step
        const double s2 = yn[0] * yn[0] + yn[1] * yn[1];
        bool far = s2 > P.range_sq[1];
        if(__builtin_expect((s2 >= P.range_sq[0]) & (s2 <= P.range_sq[1]), 0)) far = sqrt(s2) > P.range_limit;
        brk = (yn[2] > P.vert_limit) || far;
        gnd = yn[2] < P.ground;
    }
    static DEVINL void accept(RayCtx& C){}
    static DEVINL void resume(const GeoacDevParams& P, RayCtx& C, const double* y){}
    static DEVINL void restart(const GeoacDevParams& P, RayCtx& C, const double* y){}
    // GeoAc_Jacobian / GeoAc_Amplitude: 3DStratified.cpp:410-451 (nu_mag0 sign slip, c_prop[2] without w: Q4)
    static DEVINL void arrival(const GeoacDevParams& P, const RayCtx& C, int slot, const double* yn, double* R){
        double z0 = P.z_grnd < P.src[2] ? P.src[2] : P.z_grnd;
        Medium mg = medium_at(P, P.z_grnd);
        double incl = -asin(mg.c / C.c0 * yn[3]) * 180.0 / kPi;            // GeoAc3D_main.cpp:281-284
        double baz = P.phi_deg[slot] + 180.0;
        while(baz > 180.0) baz -= 360.0;
        while(baz < -180.0) baz += 360.0;
        R[GEOAC_REC_INCL] = incl;
        R[GEOAC_REC_BACKAZ] = baz;
        R[GEOAC_REC_RANGE] = sqrt(yn[0] * yn[0] + yn[1] * yn[1]);
        if(AMP){
            double amp, D;
            amp_jac(P, C, slot, yn, amp, D);
            R[GEOAC_REC_AMP] = amp;
            R[GEOAC_REC_JACOB] = D;
        }
    }
    static DEVINL void amp_jac(const GeoacDevParams& P, const RayCtx& C, int slot, const double* yn, double& amp, double& D){
        double z0 = P.z_grnd < P.src[2] ? P.src[2] : P.z_grnd;
        Medium m = medium_at(P, yn[2]);
        Medium m0 = medium_at(P, z0);
        double nx = C.a[0], ny = C.a[1], nz = yn[3];
        double nu_mag  = (m0.c - nx * m.u - ny * m.v) / m.c;
        double nu_mag0 = 1.0 - (nx * m0.u - ny * m0.v) / m0.c;
        double cp0 = m.c * nx / nu_mag + m.u, cp1 = m.c * ny / nu_mag + m.v, cp2 = m.c * nz / nu_mag;
        double cq0 = m0.c * nx / nu_mag0 + m0.u, cq1 = m0.c * ny / nu_mag0 + m0.v;
        double qx = nx / nu_mag0, qy = ny / nu_mag0;
        double cq2 = m0.c * sqrt(1.0 - qx * qx - qy * qy);
        double cpm = sqrt(cp0 * cp0 + cp1 * cp1 + cp2 * cp2);
        double cqm = sqrt(cq0 * cq0 + cq1 * cq1 + cq2 * cq2);
        double dxds = cp0 / cpm, dyds = cp1 / cpm, dzds = cp2 / cpm;
        D = dxds * (yn[5] * yn[10] - yn[9] * yn[6]) - yn[4] * (dyds * yn[10] - dzds * yn[9]) + yn[8] * (dyds * yn[6] - dzds * yn[5]);
        double th_l = P.theta_deg[slot] * kPi / 180.0;
        double num = m.rho * nu_mag * (m.c * m.c * m.c) * cqm * cos(th_l);
        double den = m0.rho * nu_mag0 * (m0.c * m0.c * m0.c) * cpm * D;
        amp = 1.0 / (4.0 * kPi) * sqrt(fabs(num / den));
    }
    // ApproximateIntercept + SetReflectionConditions: 3DStratified.cpp:136-186 (quadratic term kept)
    static DEVINL void reflect(const GeoacDevParams& P, const RayCtx& C, const double* yn, double* y, const double* ym2){
        double dz_k = yn[2] - y[2];
        double dz_g = y[2] - P.ground;
        double prev[E];
        #pragma unroll
        for(int e = 0; e < E; e++)
            prev[e] = y[e] + (y[e] - yn[e]) / dz_k * dz_g + 1.0 / 2.0 * (yn[e] + ym2[e] - 2.0 * y[e]) / (dz_k * dz_k) * (dz_g * dz_g);
        Medium mg = medium_at(P, P.z_grnd);
        double dnuz_ds = -1.0 / mg.c * (C.c0 / mg.c * mg.dc + C.a[0] * mg.du + C.a[1] * mg.dv);
        #pragma unroll
        for(int e = 0; e < E; e++) y[e] = prev[e];
        y[3] = -prev[3];
        if(AMP){
            y[6] = -prev[6]; y[10] = -prev[10];
            double den = mg.c / C.c0 * prev[3];
            y[7]  = -prev[7]  + 2.0 * dnuz_ds * prev[6]  / den;
            y[11] = -prev[11] + 2.0 * dnuz_ds * prev[10] / den;
        }
    }
    // 3DStratified.cpp:348-405 (c(0,0,0) instead of c0, w ignored: Q5) and :456-490; A, B = the segment's two path rows, aux = nu_x, nu_y of the ray
    static DEVINL void pp_aux(const GeoacDevParams& P, const double* st, size_t np, double* aux){ aux[0] = st[(ST_AUX0 + 0) * np]; aux[1] = st[(ST_AUX0 + 1) * np]; }
    static DEVINL void pp_geom(const GeoacDevParams& P, const double* aux, const double* A, const double* B, PPGeom& G, double* ref = nullptr){
        double ax = A[0], ay = A[1], az = A[2], anz = A[3];
        double dx = B[0] - ax, dy = B[1] - ay, dz = B[2] - az;
        G.ds_tt = G.ds_at = fsqrt(dx * dx + dy * dy + dz * dz);
        G.x = az + dz / 2.0;
        G.a0 = anz + (B[3] - anz) / 2.0;
        G.a1 = G.a2 = G.a3 = 0.0;
    }
    static DEVINL double pp_tt(const GeoacDevParams& P, const double* aux, const PPGeom& G, double T, double u, double v){
        double nx = aux[0], ny = aux[1];
        double qT = kGamR * T;
        double c = qT * frsq(qT);
        double cn = (c * c) * frcp(P.c000 - nx * u - ny * v);
        double cp0 = cn * nx + u, cp1 = cn * ny + v, cp2 = cn * G.a0;
        return G.ds_tt * frsq(cp0 * cp0 + cp1 * cp1 + cp2 * cp2);
    }
    static DEVINL void segment(const GeoacDevParams& P, const double* st, size_t np, const double* a, const double* b, double& tt, double& at){
        double A[4], B[4], aux[2];
        pp_aux(P, st, np, aux);
        #pragma unroll
        for(int c = 0; c < 4; c++){ A[c] = a[c * np]; B[c] = b[c * np]; }
        pp_exact<Eq3D<AMP_>>(P, aux, A, B, tt, at);
    }
};

// Two lanes per ray for the 3-D stratified set with amplitudes: lanes 2j, 2j+1 both integrate the base ray (x, y, z, nu_z) and ONE of the
// two launch-angle systems each (3DStratified.cpp:251-310 is the same code once per angle), as EqGlobalPair does for the spherical set.
struct Eq3DPair : Eq3D<true> {
    using Full = Eq3D<true>;
    static constexpr int E = 8, LANES = 2;
    static constexpr bool SPLIT = true; static constexpr bool ROW_SPLIT = true;
    template <typename TabPtr>
    static DEVINL void rhs(TabPtr tab, const GeoacDevParams& P, int& seg, const RayCtx& C, const double* y0, const double* yt, int stage, double* dy){
        cart3_rhs<true, 1>(tab, P, seg, C, yt, dy, (int)(threadIdx.x & 1));
    }
    // quadratic intercept + reflection of the base ray and of this lane's system (3DStratified.cpp:136-186)
    static DEVINL void reflect(const GeoacDevParams& P, const RayCtx& C, const double* yn, double* y, const double* ym2){
        double dz_k = yn[2] - y[2];
        double dz_g = y[2] - P.ground;
        double prev[E];
        #pragma unroll
        for(int e = 0; e < E; e++)
            prev[e] = y[e] + (y[e] - yn[e]) / dz_k * dz_g + 1.0 / 2.0 * (yn[e] + ym2[e] - 2.0 * y[e]) / (dz_k * dz_k) * (dz_g * dz_g);
        Medium mg = medium_at(P, P.z_grnd);
        double dnuz_ds = -1.0 / mg.c * (C.c0 / mg.c * mg.dc + C.a[0] * mg.du + C.a[1] * mg.dv);
        #pragma unroll
        for(int e = 0; e < E; e++) y[e] = prev[e];
        y[3] = -prev[3];
        y[6] = -prev[6];
        y[7] = -prev[7] + 2.0 * dnuz_ds * prev[6] / (mg.c / C.c0 * prev[3]);
    }
};

template <bool AMP_> struct Eq2D {
    static constexpr bool AMP = AMP_;
    static constexpr bool SEG1D = true;
    static constexpr int AUX_SAVE = 0;
    static constexpr bool COOP = false; static constexpr bool CACHE = false; static constexpr bool LDS_STATE = false; static constexpr int XCHG_BYTES = 0; static constexpr bool PP_TILE = false; static constexpr bool PP_DEDUP = false; static constexpr int SYS_SHIFT = 0; static constexpr bool ROW_SPLIT = false;
    static constexpr int NB = 3, NS = 3;
    static constexpr int PP_WAVES = 4;
    static constexpr int E = AMP_ ? 6 : 3, PW = 2, HIDX = 1, LANES = 1;
    static constexpr bool SPLIT = false;
    static constexpr bool KM2 = true, KM2_MEM = false, HMAX_PER_LEG = false;

    // GeoAc_SetInitialConditions: EquationSets.2DStratified.cpp:38-68
    static DEVINL void init(const GeoacDevParams& P, double th, double ph, double* y, RayCtx& C){
        double z0 = P.src[0] < P.z_grnd ? P.z_grnd : P.src[0];       // GeoAc2D_main.cpp:104
        Medium m = medium_at(P, z0);
        C.a[0] = cos(ph); C.a[1] = sin(ph); C.a[2] = cos(th); C.a[3] = sin(th); C.a[4] = 0.0; C.a[5] = 0.0;
        C.c0 = m.c + m.u * C.a[0] + m.v * C.a[1];
        C.nu0 = 1.0 / C.c0;
        y[0] = 0.0; y[1] = z0; y[2] = C.a[3];
        if(AMP) y[5] = C.a[2];
    }
    static DEVINL void fan_init(const GeoacDevParams& P){}
    static DEVINL double height(const GeoacDevParams& P, const double* y){ return y[1]; }
    static DEVINL double above_ground(const GeoacDevParams& P, const double* y){ return y[1] - P.ground; }
    template <typename TabPtr>
    static DEVINL void rhs(TabPtr tab, const GeoacDevParams& P, int& seg, const RayCtx& C, const double* y0, const double* yt, int stage, double* dy){
        cart2_rhs<AMP>(tab, P, seg, C, yt, dy);
    }
    // 2DStratified.cpp:194-212
    static DEVINL void checks(const GeoacDevParams& P, RayCtx& C, const double* y, const double* yn, int k, bool& brk, bool& gnd){
        brk = (yn[1] > P.vert_limit) || (yn[0] > P.range_limit);
        gnd = yn[1] < P.ground;
    }
    static DEVINL void accept(RayCtx& C){}
    static DEVINL void resume(const GeoacDevParams& P, RayCtx& C, const double* y){}
    static DEVINL void restart(const GeoacDevParams& P, RayCtx& C, const double* y){}
    // GeoAc2D_main.cpp:216-226; Jacobian / Amplitude: 2DStratified.cpp:291-313
    static DEVINL void arrival(const GeoacDevParams& P, const RayCtx& C, int slot, const double* yn, double* R){
        R[GEOAC_REC_INCL] = -P.theta_deg[slot];
        R[GEOAC_REC_BACKAZ] = 0.0;
        R[GEOAC_REC_RANGE] = yn[0];
        if(AMP){
            double amp, D;
            amp_jac(P, C, slot, yn, amp, D);
            R[GEOAC_REC_AMP] = amp;
            R[GEOAC_REC_JACOB] = D;
        }
    }
    static DEVINL void amp_jac(const GeoacDevParams& P, const RayCtx& C, int slot, const double* yn, double& amp, double& D){
        Medium m = medium_at(P, yn[1]);
        Medium mg = medium_at(P, P.z_grnd);
        double drds = m.c / C.c0 * C.a[2];
        double dzds = m.c / C.c0 * yn[2];
        D = yn[0] * (drds * yn[4] - dzds * yn[3]);
        double num = m.rho * m.c * C.a[2];
        double den = mg.rho * C.c0 * D;
        amp = 1.0 / (4.0 * kPi) * sqrt(fabs(num / den));
    }
    // 2DStratified.cpp:74-117
    static DEVINL void reflect(const GeoacDevParams& P, const RayCtx& C, const double* yn, double* y, const double* ym2){
        double dz_k = yn[1] - y[1];
        double dz_g = y[1] - P.ground;
        double prev[E];
        #pragma unroll
        for(int e = 0; e < E; e++)
            prev[e] = y[e] + (y[e] - yn[e]) / dz_k * dz_g + 1.0 / 2.0 * (yn[e] + ym2[e] - 2.0 * y[e]) / (dz_k * dz_k) * (dz_g * dz_g);
        Medium mg = medium_at(P, P.z_grnd);
        double ced = mg.dc + mg.du * C.a[0] + mg.dv * C.a[1];
        double dnuz_ds = -C.c0 / (mg.c * mg.c) * ced;
        y[0] = prev[0]; y[1] = P.ground; y[2] = -prev[2];
        if(AMP){
            y[3] = prev[3]; y[4] = -prev[4];
            y[5] = -prev[5] + 2.0 * dnuz_ds * prev[4] / (mg.c / C.c0 * prev[2]);
        }
    }
    // 2DStratified.cpp:217-286; A, B = the segment's two path rows, aux = cos / sin of the ray's azimuth
    static DEVINL void pp_aux(const GeoacDevParams& P, const double* st, size_t np, double* aux){ aux[0] = st[(ST_AUX0 + 0) * np]; aux[1] = st[(ST_AUX0 + 1) * np]; }
    static DEVINL void pp_geom(const GeoacDevParams& P, const double* aux, const double* A, const double* B, PPGeom& G, double* ref = nullptr){
        double ar = A[0], az = A[1];
        double dr = B[0] - ar, dz = B[1] - az;
        G.x = az + dz / 2.0;
        G.ds_tt = G.ds_at = fsqrt(dr * dr + dz * dz);
        G.a0 = G.a1 = G.a2 = G.a3 = 0.0;
    }
    static DEVINL double pp_tt(const GeoacDevParams& P, const double* aux, const PPGeom& G, double T, double u, double v){
        double qT = kGamR * T;
        double c = qT * frsq(qT);
        return G.ds_tt * frcp(c + u * aux[0] + v * aux[1]);
    }
    static DEVINL void segment(const GeoacDevParams& P, const double* st, size_t np, const double* a, const double* b, double& tt, double& at){
        double A[2], B[2], aux[2];
        pp_aux(P, st, np, aux);
        A[0] = a[0]; A[1] = a[np]; B[0] = b[0]; B[1] = b[np];
        pp_exact<Eq2D<AMP_>>(P, aux, A, B, tt, at);
    }
};

template <class EQ>
DEVINL void pp_exact(const GeoacDevParams& P, const double* aux, const double* A, const double* B, double& tt, double& at){
    PPGeom G;
    EQ::pp_geom(P, aux, A, B, G);
    const double xe = clampd(G.x, P.x_min, P.x_max);
    const int k = seg_guess(P.seg, P, xe);
    double T, u, v; seg_eval_f(P.seg + (size_t)k * GEOAC_SEGW, xe, T, u, v);
    tt = EQ::pp_tt(P, aux, G, T, u, v);
    const double qT = kGamR * T;
    at = suthbass_alpha(P, G.x - P.r_earth, qT * frsq(qT), rho_eval(P, k, xe), P.freq, P.T_o, P.P_o, P.cbrt_To) * G.ds_at;
}

#pragma clang fp contract(off)
#if GEOAC_AB
#include "geoac_duo.h"                // (A/B builds only, `make AB=1`: the two-wave kernel of round 3 - correct, bit-identical, measured slower)
#include "geoac_trio.h"               // (A/B builds only: the three-wave kernel of round 4 - the ray on one wave, one launch-angle system on each of two more; the same verdict)
#endif

// ------------------------------------------------------------------------------------------------
// k_init: launch angles -> initial conditions + per-ray state
// ------------------------------------------------------------------------------------------------
template <class EQ>
__global__ void __launch_bounds__(256) k_init(GeoacDevParams P){
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= P.n_pad) return;
    if(i == 0) EQ::fan_init(P);
    double* st = P.state + i;
    const size_t np = (size_t)P.n_pad;
    for(int f = 0; f < ST_NSTATE; f++) st[f * np] = 0.0;
    const int ray = P.perm ? P.perm[i] : (i < P.n_rays ? i : -1);   // slots without a ray (tail padding, quad alignment of the grid sets): finished from the start
    if(ray < 0){ st[ST_DONE * np] = 1.0; return; }
    const double th = P.theta_deg[i] * kPi / 180.0;               // GeoAcGlobal_main.cpp:244
    const double ph = kPi / 2.0 - P.phi_deg[i] * kPi / 180.0;     // :245
    double y[GEOAC_MAXE];
    for(int e = 0; e < GEOAC_MAXE; e++) y[e] = 0.0;
    RayCtx C; C.ckey = -1; C.kxy = -1;
    EQ::init(P, th, ph, y, C);
    for(int e = 0; e < GEOAC_MAXE; e++) st[(ST_Y0 + e) * np] = y[e];
    st[ST_C0 * np] = C.c0; st[ST_NU0 * np] = C.nu0;
    for(int q = 0; q < 6; q++) st[(ST_AUX0 + q) * np] = C.a[q];
    st[ST_SEG * np] = P.gtab ? -1.0 : (double)seg_guess(P.seg, P, clampd(y[EQ::HIDX], P.x_min, P.x_max));   // grid sets: search from scratch
    double* R = P.rec + (size_t)ray * (P.bounces + 1) * GEOAC_REC_STRIDE;
    for(int q = 0; q < (P.bounces + 1) * GEOAC_REC_STRIDE; q++) R[q] = 0.0;
}

// ------------------------------------------------------------------------------------------------
// k_rk4: GeoAc_Propagate_RK4 (GeoAc.Solver.cpp:12-72) for one epoch, one ray per lane
// ------------------------------------------------------------------------------------------------
// arrival row of a leg: the pair kernels hand the assembled full row to the one-lane policy's arrival()
template <class EQ, bool S = EQ::SPLIT> struct ArrivalOf { static DEVINL void go(const GeoacDevParams& P, const RayCtx& C, int slot, const double* y, double* R){ EQ::arrival(P, C, slot, y, R); } };
template <class EQ> struct ArrivalOf<EQ, true> { static DEVINL void go(const GeoacDevParams& P, const RayCtx& C, int slot, const double* y, double* R){ EQ::Full::arrival(P, C, slot, y, R); } };
template <class EQ> DEVINL void arrival_of(const GeoacDevParams& P, const RayCtx& C, int slot, const double* y, double* R){ ArrivalOf<EQ>::go(P, C, slot, y, R); }

// one path row: `p` = this lane's address of the row (component 0; k_rk4 keeps it as a running pointer, one row further per call)
template <class EQ>
DEVINL void write_row(const GeoacDevParams& P, double* p, int q, const double* y){
    if(!EQ::ROW_SPLIT && EQ::LANES > 1 && q != 0) return;   // multi-lane grid kernels: the lanes hold the same row, lane 0 stores it
    if(EQ::ROW_SPLIT){                    // pair kernels: each lane stores half of the row (Global: r, lat, lon | nu_r, nu_t, nu_p; 3D: x, y | z, nu_z)
        constexpr int H = EQ::PW / 2;
        p += (size_t)(H * q) * P.n_pad;
        #pragma unroll
        for(int c = 0; c < H; c++) p[(size_t)c * P.n_pad] = q ? y[c + H] : y[c];
    } else if(EQ::PW == 6){                // Global: r, lat, lon, nu_r, nu_t, nu_p
        #pragma unroll
        for(int c = 0; c < 6; c++) p[(size_t)c * P.n_pad] = y[c];
    } else if(EQ::PW == 4){                // 3D: x, y, z, nu_z
        #pragma unroll
        for(int c = 0; c < 4; c++) p[(size_t)c * P.n_pad] = y[c];
    } else {                               // 2D: r, z
        p[0] = y[0]; p[P.n_pad] = y[1];
    }
}

template <class EQ, bool LDS, bool SMP>
__global__ void __launch_bounds__(EQ::COOP ? 64 : 256, EQ::COOP ? GEOAC_COOP_WAVES : 1) k_rk4(GeoacDevParams P){
    constexpr int E = EQ::E;
    __builtin_amdgcn_s_setprio(3);      // latency-critical serial recurrence: win VALU arbitration against co-resident post-pass waves
    extern __shared__ double lds_tab[];
    if(threadIdx.x == 0) atomicAdd(&P.counters[5], 1ull);       // this workgroup holds its CU now: k_gate releases the previous epoch's post-pass
    // ---- sub-epochs (cooperative grid kernels, one wave per workgroup): a fan of W waves on S wave slots runs ceil(W / S) rounds per
    // epoch, the last one part empty (config-4 share: 1493 waves on 1024 slots, 27 % of the slot time idle).  With P.sub > 1 the grid is
    // sub x sub_w workgroups: workgroup (h, w) integrates rows [h, h + 1) s_rows / sub of wave w's epoch and starts from the state
    // workgroup (h - 1, w) left behind, which it waits for on a flag.  Workgroups are dispatched in index order, round robin over the
    // XCDs; sub_w is a multiple of 8, so (h - 1, w) went to the same XCD earlier and is resident or finished: the wait cannot deadlock (and
    // is bounded all the same: ~2 s, then the error flag).  The rounds of the launch are now 1 / sub as long: the idle part shrinks with it.
    const int sub_h = (EQ::COOP && P.sub > 1) ? (int)(blockIdx.x / (unsigned)P.sub_w) : 0;
    const unsigned bidx = (EQ::COOP && P.sub > 1) ? blockIdx.x % (unsigned)P.sub_w : blockIdx.x;
    const bool sub_on = EQ::COOP && P.sub > 1;
    if(EQ::COOP && sub_h > 0){
        const long long t0 = wall_clock64();
        bool ok = true;
        while(__hip_atomic_load(P.sub_flags + bidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sub_h){
            __builtin_amdgcn_s_sleep(32);
            if(wall_clock64() - t0 > (P.sub_test_stall ? 2000000ll : 200000000ll)){ ok = false; break; }          // (100 MHz counter: 2 s)
        }
        if(!ok){ if(threadIdx.x == 0) atomicOr(&P.counters[2], 4ull); return; }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                      // (invalidates this CU's L1: the state rows another CU stored)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int row_end = sub_on ? (int)(((long long)P.s_rows * (sub_h + 1)) / P.sub) : P.s_rows;
    const int tid0 = bidx * blockDim.x + threadIdx.x;
    const int tid = (P.spread > 1) ? tid0 / P.spread : tid0;
    // `col` = this ray's column in the epoch's chunk buffers (path, contrib, row counts, leg-end and sample events); `slot` = its row in the
    // fan-long state block.  Without compaction they are the same number.  With it (P.colmap) the epoch runs over the dense list of the
    // rays that were still alive after the previous epoch (k_compact): column p integrates slot colmap[p], finished rays hold no lane.
    const int col = P.slot_lo + tid / EQ::LANES;
    const int q = tid % EQ::LANES;                              // pair kernel: which derivative system this lane carries; quad kernels: which cell corner
    const int qs = EQ::SPLIT ? (q >> EQ::SYS_SHIFT) : 0;        // split-state policies: the launch-angle system of this lane (eight-lane kernel: q >> 2)
    const int col_hi = P.colmap ? min(P.slot_hi, *P.n_cols) : P.slot_hi;
    const bool mine = !(P.spread > 1 && (tid0 & (P.spread - 1))) && col < col_hi;   // spread > 1: sparse lanes (grid sets, small fans)
    const int slot = (P.colmap && mine) ? P.colmap[col] : col;  // ray slot
    const size_t np = (size_t)P.n_pad;
    double* st = P.state + (mine ? slot : 0);
    bool done = mine ? (st[ST_DONE * np] != 0.0) : true;
    if(mine && done && sub_h == 0){ P.nrows[col] = 0; P.nlegend[col] = 0; if(SMP) P.nev[col] = 0; }
    // a workgroup whose rays have all finished leaves before the table is staged (late epochs, and the launch the host
    // enqueues ahead of knowing that the previous epoch finished the fan)
    if(!__syncthreads_or(!done)){
        if(EQ::COOP && sub_on && threadIdx.x == 0) __hip_atomic_store(P.sub_flags + bidx, sub_h + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (nothing stored: nothing to release)
        return;
    }
    // ---- stage the segment table in LDS (coalesced 8 B/lane loads; 153 KiB for ToyAtmo) ----
    const double* gtab = P.seg;
    if(LDS){
        const int total = P.nseg * GEOAC_SEGW;
        for(int q = threadIdx.x; q < total; q += blockDim.x) lds_tab[q] = gtab[q];
        __syncthreads();
    }
    // cooperative-gather policies keep every lane of the wave in the step loop: a lane whose ray has finished (or a padding lane) goes on
    // fetching table records for its quad mates (grid_eval3_coop); all its own side effects are switched off
    const bool idle0 = done;                                     // finished before this epoch: nothing of this lane's is written
    // record-cache kernels (one wave per workgroup): behind the state rows, 64 x 976 B of per-lane records, then a copy of the z nodes,
    // staged by ALL 64 lanes before the lanes without a live ray leave
    constexpr int LDS_STATE_BYTES = (EQ::COOP ? EQ::XCHG_BYTES : 0) + (EQ::LDS_STATE ? 2 * E * 64 * (int)sizeof(double) : 0);      // (rows of E components: an amplitude-less kernel's 6 leave room for a second wave per CU beside the record cache)
    char* const ldsc = (char*)lds_tab + (threadIdx.x >> 6) * LDS_STATE_BYTES + LDS_STATE_BYTES;
    if(EQ::CACHE){
        double* gzl = (double*)(ldsc + GEOAC_CACHE_BYTES);
        for(int i = threadIdx.x & 63; i <= P.nseg; i += 64) gzl[i] = P.gz[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");          // every lane reads entries other lanes wrote
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    }
    if(!EQ::COOP && done) return;
    // per wave: (COOP) the 64 x 176 B exchange buffer of the cooperative gather, then (LDS_STATE) y[E][64] and yn[E][64]
    constexpr int LDS_XCHG_BYTES = EQ::COOP ? EQ::XCHG_BYTES : 0;
    constexpr int LDS_WAVE_BYTES = LDS_XCHG_BYTES + (EQ::LDS_STATE ? 2 * E * 64 * (int)sizeof(double) : 0);
    char* const ldsw = (char*)lds_tab + (threadIdx.x >> 6) * LDS_WAVE_BYTES;
    double* const ly = (double*)(ldsw + LDS_XCHG_BYTES) + (threadIdx.x & 63);
    double* const lyn = ly + E * 64;

    int nr = 0, nle = 0;
    if(EQ::COOP && sub_h > 0 && !idle0){ nr = P.nrows[col]; nle = P.nlegend[col]; }   // continue the chunk where the previous sub-epoch stopped
    static_assert(E <= ST_K - ST_Y0, "state block: y[] overruns ST_K");
    static_assert(!EQ::KM2 || ST_YM2 + E <= ST_DPREV, "state block: ym2[] overruns ST_DPREV");
    constexpr bool YM2_REG = EQ::KM2 && !EQ::KM2_MEM;                // row k-2 in registers (else: in the state block, read at reflections)
    double y[E], ym2[YM2_REG ? E : 1];
    #pragma unroll
    for(int e = 0; e < E; e++) y[e] = st[(ST_Y0 + ((EQ::SPLIT && e >= EQ::NB) ? e + EQ::NS * qs : e)) * np];
    if(YM2_REG){
        #pragma unroll
        for(int e = 0; e < E; e++) ym2[e] = st[(ST_YM2 + ((EQ::SPLIT && e >= EQ::NB) ? e + EQ::NS * qs : e)) * np];
    }
    int k = (int)st[ST_K * np];                                   // steps of the current leg (step_limit is clamped to 2^31 - 1 on the host)
    const int k_lim = (int)(P.step_limit - 1);
    int leg = (int)st[ST_LEG * np];
    double hmax = st[ST_HMAX * np];
    RayCtx C;
    C.ckey = -1; C.kxy = -1;
    C.c0 = st[ST_C0 * np]; C.nu0 = st[ST_NU0 * np];
    #pragma unroll
    for(int q = 0; q < 6; q++) C.a[q] = st[(ST_AUX0 + q) * np];
    EQ::resume(P, C, y);
    int seg = P.gtab ? (int)st[ST_SEG * np] : (int)st[ST_SEG * np] * GEOAC_SEGW;   // 1-D sets: element offset of the current spline segment; grid sets: vertical segment index
    if constexpr (EQ::SEG1D){ if(LDS) seg_fetch(lds_tab, seg, C.rec); else seg_fetch(gtab, seg, C.rec); }
    unsigned long long steps_here = 0;
    int nev = (EQ::COOP && SMP && sub_h > 0 && !idle0) ? P.nev[col] : 0;   // WriteRays / WriteCaustics events of this chunk
    double dprev = SMP ? st[ST_DPREV * np] : 0.0;               // Jacobian of the previous row (caustic detection)
    const bool want_rays = SMP && (P.mode & GEOAC_MODE_WRITE_RAYS), want_caus = SMP && EQ::AMP && (P.mode & GEOAC_MODE_WRITE_CAUSTICS);

    // this lane's address of chunk row nr: a running pointer in the 1-D kernels (kept in step with nr), recomputed per row in the grid kernels
    // (two registers they do not have)
    const size_t row_stride = (size_t)EQ::PW * np;
    double* prow = P.path + (size_t)nr * row_stride + col;
    auto put_row = [&](const double* v){
        if(EQ::SEG1D){ write_row<EQ>(P, prow, q, v); prow += row_stride; }
        else write_row<EQ>(P, P.path + (size_t)nr * row_stride + col, q, v);
        nr++;
    };
    if(!EQ::COOP || (!idle0 && sub_h == 0)) put_row(y);          // carry row: chunk row 0 = current state

    // ---- what opens a step on row ya: running turning height, raypath / caustic events of the row ----
    auto open_step = [&](const double* ya){
        // running turning height: max over rows m < k of the height component   (GeoAcGlobal_main.cpp:294)
        if(EQ::PW == 2 && (P.mode & GEOAC_MODE_INTERACTIVE)){         // GeoAc2D -interactive: max over rows 1..k-1 of solution[m][2] (nu_z)
            if(k >= 1) hmax = (hmax < ya[2]) ? ya[2] : hmax;
        } else {
            double h = EQ::height(P, ya); if(EQ::HMAX_PER_LEG && k == 0) hmax = 0.0;
            if(EQ::SEG1D) hmax = __builtin_fmax(hmax, h);         // (one v_max_f64 instead of a compare and two selects; no NaN reaches it: the same value)
            else hmax = (hmax < h) ? h : hmax;                    // (grid kernels: the select form they were tuned with)
        }

        if(SMP && k >= 1){
            // ya is row m = k (1 <= m < k_final) at chunk row nr-1: the rows the reference's post-pass loop visits
            // (GeoAcGlobal_main.cpp:264-286).  Raypath sample every sample_stride-th row; caustic where the Jacobian changes sign.
            const bool smp_row = want_rays && (k % P.smp_stride == 0);
            if(smp_row || want_caus){
                double amp = 0.0, D = 0.0;
                if(EQ::AMP) EQ::amp_jac(P, C, slot, ya, amp, D);
                if(smp_row){
                    if(nev < P.ev_cap){ P.ev_row[(size_t)nev * np + col] = nr - 1; P.ev_m[(size_t)nev * np + col] = (int)k; P.ev_amp[(size_t)nev * np + col] = amp; }
                    nev++;
                }
                if(want_caus){
                    if(k > 1 && D * dprev < 0.0){
                        if(nev < P.ev_cap){ P.ev_row[(size_t)nev * np + col] = nr - 1; P.ev_m[(size_t)nev * np + col] = (int)k | (1 << 30); P.ev_amp[(size_t)nev * np + col] = 0.0; }
                        nev++;
                    }
                    dprev = D;
                }
            }
        }
    };
    // ---- what closes it: ya = row k-1, yb = the new row k.  The row is stored and tested (GeoAc_BreakCheck / GeoAc_GroundCheck / the loop bound);
    //      true: it ended the leg ----
    bool brk = false, gnd = false, lim = false;
    auto advance = [&](const double* ya, const double* yb) -> bool {
        k++; steps_here++;
        put_row(yb);
        EQ::checks(P, C, ya, yb, k, brk, gnd);
        lim = (k >= k_lim);                                       // Solver.cpp loop bound; never reached on sane inputs
        return brk || gnd || lim;
    };
    // ---- leg end: ya = row k-1, yb = the leg's last row k, yc = row k-2 (YM2_REG).  The new leg's row 0 (for a finished ray: row k-1) is left in ya ----
    auto leg_end = [&](double* ya, double* yb, const double* yc){
        // ---- leg end: record (GeoAcGlobal_main.cpp:293-317 and twins) ----
        double* R = P.rec + ((size_t)(P.perm ? P.perm[slot] : slot) * (P.bounces + 1) + leg) * GEOAC_REC_STRIDE;
        R[GEOAC_REC_STEPS] = (double)((lim && !brk && !gnd) ? k + 1 : k);   // exhausted loop: the reference returns step_limit (= k + 1), Solver.cpp:70
        P.legend[(size_t)nle * np + col] = nr - 1; nle++;
        if(lim && !brk && !gnd){ atomicOr(&P.counters[2], 1ull); steps_here++; }   // (the step total is the sum of the reference's return values: step_limit for this leg)
        // the leg's last row solution[k][*] (both outcomes: arrival rows read it, and so do the eigenray scans' messages after a break)
        double yf[18];
        if(EQ::SPLIT){
            // assemble the reference's full row (base + both systems) from the lane pair (both lanes end up with the same row
            // and store the same record)
            #pragma unroll
            for(int e = 0; e < 18; e++) yf[e] = 0.0;
            #pragma unroll
            for(int e = 0; e < EQ::NB; e++) yf[e] = yb[e];
            #pragma unroll
            for(int e = 0; e < EQ::NS; e++){
                double mine = yb[EQ::NB + e], other = __shfl_xor(mine, 1 << EQ::SYS_SHIFT);
                yf[EQ::NB + e]  = qs ? other : mine;
                yf[EQ::NB + EQ::NS + e] = qs ? mine : other;
            }
        } else {
            #pragma unroll
            for(int e = 0; e < 18; e++) yf[e] = (e < E) ? yb[e < E ? e : 0] : 0.0;
        }
        #pragma unroll
        for(int e = 0; e < (EQ::SPLIT ? EQ::NB + 2 * EQ::NS : E); e++) R[GEOAC_REC_STATE + e] = yf[e];
        if(brk){
            R[GEOAC_REC_BROKE] = 1.0;
            done = true;
        } else {
            R[GEOAC_REC_VALID] = 1.0;
            if(lim && !gnd){                                  // exhausted loop: the reference's row count is one more, its turning height covers the last row too
                const double hl = EQ::height(P, yb);
                if(!(EQ::PW == 2 && (P.mode & GEOAC_MODE_INTERACTIVE))) hmax = (hmax < hl) ? hl : hmax;
            }
            R[GEOAC_REC_TURN] = hmax;
            // the arrival part of the record (inclination, back azimuth, range, amplitude, Jacobian: sin / cos / asin / atan2 and - grid sets - table
            // evaluations at the leg's last row, which the record holds) is filled in by k_arrival once the fan has finished: thousands of
            // instructions and the constants hipcc hoists out of them stay out of this kernel's registers
            if(leg >= P.bounces){
                done = true;
            } else {
                if(EQ::KM2 && EQ::KM2_MEM){
                    double r2[E];
                    #pragma unroll
                    for(int e = 0; e < E; e++) r2[e] = st[(ST_YM2 + ((EQ::SPLIT && e >= EQ::NB) ? e + EQ::NS * qs : e)) * np];
                    EQ::reflect(P, C, yb, ya, r2);
                } else EQ::reflect(P, C, yb, ya, yc);
                leg++; k = 0;
                EQ::restart(P, C, ya);
                put_row(ya);                                      // leg-start row
            }
        }
    };

    if constexpr (EQ::SEG1D){
        // ---- stratified sets: the serial chain of one ray IS the run time (one wave per SIMD issues an instruction every ~4.5 cycles whatever its
        // kind), so the loop is laid out for instruction count:
        //  * stages 0 and 3 are peeled off the rolled stage loop (four copies of the right-hand side do not fit the registers, three do): stage 0
        //    reads the row itself (no copy, no rotation of the carried sin / cos), stage 3 forms no further stage input;
        //  * the new row is written over the old one, component by component, once the tests on its position have passed - no copy of the row at
        //    the bottom of the loop;
        //  * a row that ends a leg (or fills a lane's chunk) takes the WHOLE wave out of the loop before that (a vote: a uniform branch), with the row still in its parts
        //    (row k-1, the three-stage sum, the last slope); the leg end is worked off outside and the wave re-enters.  Inside the loop a row is
        //    never merged with the outcome of a rare path, which is what cost ~45 register moves per step before (a few hundred leg ends per
        //    wave and fan against 50 000 steps).
        while(__any((nr + 2 <= P.s_rows) && !done)){                  // (wave-uniform)
            bool ev = false, pend = false;
            double dy[E], ys[E], w6 = 0.0;                            // at a vote: the last slope, the sum of the first three stages, ds / 6
            if((nr + 2 <= P.s_rows) && !done) for(;;){                // (every lane that enters leaves at the same vote: no exec-mask bookkeeping inside)
                open_step(y);
                const double ds = set_ds(EQ::above_ground(P, y), P.ds_min, P.ds_max);        // GeoAc_Set_ds (Global.cpp:210-217 and twins)
                const double ds_2 = 0.5 * ds, ds_6 = (1.0 / 6.0) * ds, ds_3 = (1.0 / 3.0) * ds;
                // k_s = ds f(y + a_s k_{s-1}), a = {0, 1/2, 1/2, 1};  y' = y + k1/6 + k2/3 + k3/3 + k4/6  (Solver.cpp:33-54)
                double yt[E];
                if(LDS) EQ::rhs(lds_tab, P, seg, C, y, y, 0, dy); else EQ::rhs(gtab, P, seg, C, y, y, 0, dy);
                #pragma unroll
                for(int e = 0; e < E; e++){ ys[e] = __builtin_fma(dy[e], ds_6, y[e]); yt[e] = __builtin_fma(dy[e], ds_2, y[e]); }
                #pragma unroll 1
                for(int stage = 1; stage < 3; stage++){
                    if(LDS) EQ::rhs(lds_tab, P, seg, C, y, yt, stage, dy); else EQ::rhs(gtab, P, seg, C, y, yt, stage, dy);
                    const double wa = (stage == 2) ? ds : ds_2;
                    #pragma unroll
                    for(int e = 0; e < E; e++){ ys[e] = __builtin_fma(dy[e], ds_3, ys[e]); yt[e] = __builtin_fma(dy[e], wa, y[e]); }
                }
                if(LDS) EQ::rhs(lds_tab, P, seg, C, y, yt, 3, dy); else EQ::rhs(gtab, P, seg, C, y, yt, 3, dy);      // (the last slope stays in dy)
                // the position part of the new row and the tests on it (GeoAc_BreakCheck / GeoAc_GroundCheck / the loop bound)
                double t[EQ::NB];
                #pragma unroll
                for(int e = 0; e < EQ::NB; e++) t[e] = __builtin_fma(dy[e], ds_6, ys[e]);
                EQ::checks(P, C, y, t, k + 1, brk, gnd);
                lim = (k + 1 >= k_lim);                               // Solver.cpp loop bound; never reached on sane inputs
                ev = brk || gnd || lim;
                const bool full = !(nr + 3 <= P.s_rows);                // this lane's chunk has no room for another step after this one
                if(__builtin_expect(__any(ev || full), 0)){ pend = true; w6 = ds_6; break; }
                #pragma unroll
                for(int e = 0; e < E; e++){
                    if(YM2_REG) ym2[e] = y[e];
                    y[e] = __builtin_fma(dy[e], ds_6, ys[e]);
                }
                k++; steps_here++;
                put_row(y);
                EQ::accept(C);
            }
            if(pend){                                                 // (rare) the row that was voted on
                #pragma unroll
                for(int e = 0; e < E; e++) ys[e] = __builtin_fma(dy[e], w6, ys[e]);
                k++; steps_here++;
                put_row(ys);
                EQ::accept(C);                                        // (a leg end sets the carried values anew: restart)
                if(ev) leg_end(y, ys, ym2);
                else {
                    #pragma unroll
                    for(int e = 0; e < E; e++){ if(YM2_REG) ym2[e] = y[e]; y[e] = ys[e]; }
                }
            }
        }
    } else {
    // COOP: wave-uniform loop (every lane stays while any lane of the wave has work; `act` predicates this lane's own work).
    // Other policies: the plain per-lane loop
    double yn[E];
    while(EQ::COOP ? (bool)__any((nr + 2 <= row_end) && !done) : ((nr + 2 <= P.s_rows) && !done)){
        const bool act = EQ::COOP ? ((nr + 2 <= row_end) && !done) : true;
#ifdef GEOAC_KSTAT
        C.ckey = (C.ckey & ~1) | (act ? 1 : 0);                 // (bit 0: live; the rest: the lane's key of the stage before, rngdep_rhs)
#endif
        double ds = P.ds_min;
        if(!EQ::COOP || act){
            open_step(y);
            // ---- GeoAc_Set_ds (Global.cpp:210-217 and twins); the form the grid kernels were tuned with (registers) ----
            ds = 0.05 - 0.049 * exp(-EQ::above_ground(P, y) / 0.75);
            ds = (P.ds_max < ds) ? P.ds_max : ds;
            ds = (ds < P.ds_min) ? P.ds_min : ds;
        }

        // ---- the four RK4 stages as ONE rolled loop (a single copy of the RHS keeps the live set < 256 VGPRs):
        //      k_s = ds f(y + a_s k_{s-1}), a = {0, 1/2, 1/2, 1};  y' = y + k1/6 + k2/3 + k3/3 + k4/6  (Solver.cpp:33-54)
        double dy[E], yt[E];
        #pragma unroll
        for(int e = 0; e < E; e++){ yt[e] = y[e]; yn[e] = y[e]; }
        if(EQ::LDS_STATE){
            // the step's base row y and the accumulating new row yn live in LDS ([component][lane], conflict free) while the four stages
            // run: 72 registers less under the table evaluation, which is what spilled (416 B of scratch per lane before)
            #pragma unroll
            for(int e = 0; e < E; e++){ ly[e * 64] = y[e]; lyn[e * 64] = y[e]; }
        }
        #pragma unroll 1
        for(int stage = 0; stage < 4; stage++){
            if(EQ::COOP) EQ::rhs((double*)ldsw, P, seg, C, y, yt, stage, dy);
            else if(EQ::CACHE) EQ::rhs((double*)ldsc, P, seg, C, y, yt, stage, dy);
            else if(LDS) EQ::rhs(lds_tab, P, seg, C, y, yt, stage, dy); else EQ::rhs(gtab, P, seg, C, y, yt, stage, dy);
            // (formed per stage - three doubles less to keep alive under the table evaluation)
            const double wa = ((stage == 2) ? 1.0 : 0.5) * ds;
            const double wb = ((stage == 0 || stage == 3) ? (1.0 / 6.0) : (1.0 / 3.0)) * ds;
            if(EQ::LDS_STATE){
                #pragma unroll
                for(int e = 0; e < E; e++){
                    lyn[e * 64] = __builtin_fma(dy[e], wb, lyn[e * 64]);
                    yt[e] = __builtin_fma(dy[e], wa, ly[e * 64]);
                }
            } else {
                #pragma unroll
                for(int e = 0; e < E; e++){
                    yn[e] = __builtin_fma(dy[e], wb, yn[e]);
                    yt[e] = __builtin_fma(dy[e], wa, y[e]);
                }
            }
        }
        if(EQ::LDS_STATE){
            #pragma unroll
            for(int e = 0; e < E; e++){ y[e] = ly[e * 64]; yn[e] = lyn[e * 64]; }
        }

        if(!EQ::COOP || act){                                     // (a helper lane has nothing of its own to advance)
            if(advance(y, yn)) leg_end(y, yn, ym2);
            else {
                if(YM2_REG){
                    #pragma unroll
                    for(int e = 0; e < E; e++) ym2[e] = y[e];
                } else if(EQ::KM2 && (EQ::LANES == 1 || (EQ::SPLIT ? (q & ((1 << EQ::SYS_SHIFT) - 1)) == 0 : q == 0))){
                    // KM2_MEM: one coalesced row per step, stored by the ray's first lane (split state: by the first lane of each launch-angle system, its part)
                    #pragma unroll
                    for(int e = 0; e < E; e++) st[(ST_YM2 + ((EQ::SPLIT && e >= EQ::NB) ? e + EQ::NS * qs : e)) * np] = y[e];
                }
                #pragma unroll
                for(int e = 0; e < E; e++) y[e] = yn[e];
                EQ::accept(C);
            }
        }
    }
    }

    // ---- save state (pair kernel: both lanes store the identical base ray; each stores its own derivative system) ----
    if(!(EQ::COOP && idle0)){
    #pragma unroll
    for(int e = 0; e < E; e++) st[(ST_Y0 + ((EQ::SPLIT && e >= EQ::NB) ? e + EQ::NS * qs : e)) * np] = y[e];
    if(YM2_REG){
        #pragma unroll
        for(int e = 0; e < E; e++) st[(ST_YM2 + ((EQ::SPLIT && e >= EQ::NB) ? e + EQ::NS * qs : e)) * np] = ym2[e];
    }
    st[ST_K * np] = (double)k; st[ST_LEG * np] = (double)leg; st[ST_DONE * np] = done ? 1.0 : 0.0;
    st[ST_HMAX * np] = hmax; st[ST_SEG * np] = P.gtab ? (double)seg : (double)(seg / GEOAC_SEGW);
    #pragma unroll
    for(int q = 0; q < EQ::AUX_SAVE; q++) st[(ST_AUX0 + q) * np] = C.a[q];      // (only what the kernel changes: a value that is merely carried would sit in registers from entry to exit)
    P.nrows[col] = nr; P.nlegend[col] = nle;
    if(SMP){
        st[ST_DPREV * np] = dprev;
        P.nev[col] = nev < P.ev_cap ? nev : P.ev_cap;
        if(nev > P.ev_cap) atomicOr(&P.counters[2], 2ull);
    }
    }

    // ---- step count and live-ray count: lanes of finished rays have already returned, so reduce over the
    //      lanes that are still here (ballot of the active mask), one atomic pair per wave ----
    if(q != 0) steps_here = 0;                                   // pair kernel: count each ray once
    const unsigned long long act = __ballot(1);
    unsigned long long s = 0;
    for(int l = 0; l < 64; l++){
        unsigned long long v = __shfl(steps_here, l);
        if((act >> l) & 1ull) s += v;
    }
    const unsigned long long live = __popcll(__ballot(!done && q == 0));
    const bool last_sub = !sub_on || sub_h == P.sub - 1;          // the live counts of the epoch are those after its last sub-epoch
    if((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1){
        atomicAdd(&P.counters[0], s);
        if(last_sub){
            atomicAdd(&P.counters[P.live_slot], live);
            if(live) atomicAdd(&P.counters[P.live_slot == 1 ? 4 : 7], 1ull);     // waves that still carry a live ray
        }
    }
    if(EQ::COOP && sub_on){
        // publish the state this workgroup stored (one wave per workgroup: its stores are ordered before the release by the fence)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // (hipcc may drop the wait behind the write-back: MI355X_MICROARCH.md, inter-workgroup visibility)
        if(threadIdx.x == 0 && !(P.sub_test_stall && sub_h == 0)) __hip_atomic_store(P.sub_flags + bidx, sub_h + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------
// k_postpass: one thread per path segment (row i -> row i+1 of one ray)
// ------------------------------------------------------------------------------------------------
// c, u, v, rho (WANT_UV = false: c and rho) at this lane's point, table read once per distinct key of the wave (k_postpass).  Every lane of
// the wave takes part; lanes without a segment carry key 0xffffffff.  Up to GEOAC_PP_SLOTS keys per round; more keys, more rounds.
template <bool GLB, bool WANT_UV>
DEVINL void pp_medium_by_key(const GeoacDevParams& P, char* wl, bool valid, unsigned key, const GridLoc& L, Medium3& out){
    const unsigned lane = threadIdx.x & 63u;
    const unsigned nn = (unsigned)(P.gnx * P.gny);
    unsigned long long todo = __ballot(valid);
    out.c = out.u = out.v = out.rho = out.dcz = out.duz = out.dvz = 0.0;
    while(todo){                                                              // (wave-uniform)
        unsigned long long rem = todo, served = 0;
        int myslot = -1;
        #pragma unroll 1
        for(int sl = 0; sl < GEOAC_PP_SLOTS && rem; sl++){
            const int l0 = __ffsll((long long)rem) - 1;
            const unsigned k0 = (unsigned)__builtin_amdgcn_readlane((int)key, l0);
            const unsigned long long m = __ballot(key == k0) & rem;
            if((m >> lane) & 1ull) myslot = sl;
            const unsigned kz0 = k0 / nn, n00 = k0 - kz0 * nn;
            char* dst = wl + sl * GEOAC_PP_SLOTB;
            #pragma unroll
            for(int j = 0; j < 2; j++){
                const unsigned id = 64u * j + lane, rec = id >> 3, c = id & 7u, field = rec >> 2, cn = rec & 3u;
                const unsigned node = n00 + (cn >> 1) * (unsigned)P.gny + (cn & 1u);
                const double* src = (field < 3u) ? P.gtab + (((size_t)field * P.nseg + kz0) * nn + node) * GRec<GLB>::N + 2 * c
                                                 : P.gtab + (size_t)3 * P.nseg * nn * GRec<GLB>::N + ((size_t)kz0 * nn + node) * GEOAC_GREC_RHO + 2 * c;
                *(geoac_d2*)(dst + 16 * id) = *(const geoac_d2*)src;
            }
            served |= m; rem &= ~m;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        if(myslot >= 0){
            const char* sb = wl + myslot * GEOAC_PP_SLOTB;
            out.c = sqrt(kGamR * grid_eval_f_slot<GLB>(L, sb));
            if(WANT_UV){ out.u = grid_eval_f_slot<GLB>(L, sb + 512); out.v = grid_eval_f_slot<GLB>(L, sb + 1024); }
            out.rho = grid_eval_f_slot<GLB>(L, sb + 1536);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");        // the next round's stores stay behind these reads
        __builtin_amdgcn_wave_barrier();
        todo &= ~served;
    }
}

template <class EQ>
__global__ void __launch_bounds__(256, EQ::PP_WAVES) k_postpass(GeoacDevParams P, int rows){
    // grid-stride sweep over (segment row i, ray slot); by default the grid covers the sweep in one pass
    // Grid sets (PP_TILE): a workgroup takes a tile of 16 rows x 16 rays, a wave 16 consecutive rows of 4 rays.  The segment midpoints of
    // one ray's consecutive rows lie in the same cell and vertical segment nearly always, so the 64 lanes of a table gather touch a handful
    // of cache lines instead of 64 (the texture path is busy 48 cycles per load with lane-private lines, 17-23 with shared ones); the path
    // rows are read in 32-byte pieces (16 lines per load) that the four waves of the workgroup share.
    const size_t np = (size_t)P.n_pad;
    const int bx = EQ::PP_TILE ? (P.n_cols_bound + 15) / 16 : (P.n_cols_bound + 255) / 256;      // column-blocks per row (tile row)
    const long long total = (long long)bx * (EQ::PP_TILE ? (rows - 1 + 15) / 16 : rows - 1);
    for(long long w = blockIdx.x; w < total; w += gridDim.x){
        const int i = EQ::PP_TILE ? (int)(w / bx) * 16 + (int)((threadIdx.x & 63u) >> 2) : (int)(w / bx);
        const int col = EQ::PP_TILE ? (int)(w % bx) * 16 + (int)((threadIdx.x >> 6) << 2) + (int)(threadIdx.x & 3u) : (int)(w % bx) * 256 + threadIdx.x;
        if constexpr (EQ::PP_DEDUP){
            // Grid sets: the 64 segments of a wave (16 rows of 4 rays) lie in a handful of (cell, vertical segment) KEYS.  The wave fetches each
            // distinct key's 16 half-records (T, u, v, rho x 4 corners x 128 B) ONCE - two fully coalesced loads - into an LDS slot, and every
            // lane evaluates its medium from the slot of its key (pp_medium_by_key): 2 K loads per wave instead of 128 lane-private ones (the
            // kernel was bound by the texture path: 95 % busy).  The spherical set evaluates a second point per segment, the reference state
            // of the absorption at ground level under the midpoint: same cell, lowest segment - one or two keys per wave.
            extern __shared__ double pp_lds[];
            char* const wl = (char*)pp_lds + (threadIdx.x >> 6) * (GEOAC_PP_SLOTS * GEOAC_PP_SLOTB);
            const bool valid = (i < rows - 1) && col < (P.colmap ? *P.n_cols : P.n_pad) && (i + 1 < P.nrows[col < (int)np ? col : 0]);
            const double* a = P.path + ((size_t)(valid ? i : 0) * EQ::PW) * np + (valid ? col : 0);
            const double* b = a + (size_t)EQ::PW * np;
            typename EQ::SegGeom G;
            GridLoc L, L2;
            unsigned key = 0xffffffffu, key2 = 0xffffffffu;
            const unsigned nn = (unsigned)(P.gnx * P.gny);
            if(valid){
                EQ::seg_mid(P, np, a, b, G);
                const double xe = clampd(G.x, P.g_lo[0], P.g_hi[0]), ye = clampd(G.y, P.g_lo[1], P.g_hi[1]), ze = clampd(G.z, P.x_min, P.x_max);
                grid_locate(P, xe, ye, ze, -1, L);
                key = (unsigned)L.kz * nn + (unsigned)L.n00;
                if(EQ::PP_TWO){
                    grid_locate(P, xe, ye, clampd(P.z_grnd, P.x_min, P.x_max), -1, L2);
                    key2 = (unsigned)L2.kz * nn + (unsigned)L2.n00;
                }
            }
            Medium3 m, g;
            pp_medium_by_key<EQ::GLOBAL, true>(P, wl, valid, key, L, m);
            if(EQ::PP_TWO) pp_medium_by_key<EQ::GLOBAL, false>(P, wl, valid, key2, L2, g);
            if(valid){
                double tt, at;
                EQ::seg_sums(P, G, m, g, tt, at);
                double* o = P.contrib + ((size_t)i * 2) * np + col;
                o[0]  = tt;
                o[np] = at;
            }
            continue;
        }
        if(EQ::PP_TILE && i >= rows - 1) continue;
        if(col >= (P.colmap ? *P.n_cols : P.n_pad)) continue;
        if(i + 1 >= P.nrows[col]) continue;
        const int slot = P.colmap ? P.colmap[col] : col;
        const double* a = P.path + ((size_t)i * EQ::PW) * np + col;
        const double* b = a + (size_t)EQ::PW * np;
        double* o = P.contrib + ((size_t)i * 2) * np + col;
        double tt, at;
        EQ::segment(P, P.state + slot, np, a, b, tt, at);
        o[0]  = tt;
        o[np] = at;
    }
}

// k_postpass_tab: the post-pass of the stratified sets with the absorption table.  One thread walks GEOAC_PP_ROWS consecutive segments
// of one ray: every path row is read once (the next one is already on its way while a segment is evaluated), and the spline record and the
// table entry of the midpoint stay in registers from one segment to the next - consecutive steps of a ray are 1 - 50 m apart, the nodes
// ~100 m - so the usual segment has no dependent memory access at all: geometry, one spline evaluation, three degree-5 polynomials and
// two square roots (~200 instructions per 64 segments instead of ~1 500).
// (compiled for two waves per SIMD for the spherical set - 172 registers; at 168 it spills two -, four for the Cartesian ones; the exact fall-back lives in k_ppfix:
//  inlined here it took the kernel to 256 + 12 registers, one wave per SIMD)
#ifndef GEOAC_PPTAB_GLOBAL_WAVES
#define GEOAC_PPTAB_GLOBAL_WAVES 2
#endif
// TBL (the spherical set on fans that fill the chip): the table entry in hand lives in LDS, [coefficient][thread] (19 x 256 doubles = 38 KiB per
// workgroup, conflict-free), not in 38 registers, and no row is prefetched - with it the kernel fits 128 registers: FOUR waves per SIMD, whose
// loads cover one another's latency (config 3's post-pass waited in 53 % of its wave cycles at two waves per SIMD).  Same operations on the
// same operands: same bits.
template <class EQ, bool ONETRIP, bool TBL = false>
__global__ void __launch_bounds__(256, (EQ::PW == 6 && !TBL) ? GEOAC_PPTAB_GLOBAL_WAVES : 4) k_postpass_tab(GeoacDevParams P, int rows, int gy0){
    constexpr int PW = EQ::PW, R = GEOAC_PP_ROWS;
    extern __shared__ double pp_tb_lds[];
    // TBL: this thread's entry, coefficient c at ltb[c * 256].  The thread index is kept as ONE register, its byte offset into that array (opaque to the
    // compiler, which would hold the index and the offset otherwise - one register too many for four waves per SIMD)
    unsigned toff = threadIdx.x << 3;
    if(TBL) asm volatile("" : "+v"(toff));
    double* const ltb = (double*)((char*)pp_tb_lds + toff);
    const int tix = TBL ? (int)(toff >> 3) : (int)threadIdx.x;
    const size_t np = (size_t)P.n_pad;
    const int ncol = P.colmap ? *P.n_cols : P.n_pad;
    const int col = (int)blockIdx.x * 256 + tix;
    if(col >= ncol) return;
    const int nr = P.nrows[col];
    const int slot = P.colmap ? P.colmap[col] : col;
    double aux[2] = { 0.0, 0.0 };
    EQ::pp_aux(P, P.state + slot, np, aux);
    {   // grid: x = blocks of 256 columns, y = groups of R rows, from group gy0 on (the host cuts a launch at the grid's y limit)
        const int i0 = (gy0 + (int)blockIdx.y) * R;
        if(i0 + 1 >= nr) return;
        const int i1 = min(i0 + R, nr - 1);                       // segments i0 .. i1 - 1
        const double* a = P.path + ((size_t)i0 * PW) * np + col;
        // rows i, i + 1 and the prefetched i + 2.  (Three buffers in rotation - three copies of the body, no row copies - were measured in round 3:
        // 12 moves fewer per segment, but the exact fall-back inlined three times takes the kernel from 164 to 268 registers, one wave per SIMD.)
        double A[PW], B[PW], Bn[TBL ? 1 : PW];
        #pragma unroll
        for(int c = 0; c < PW; c++){ A[c] = a[(size_t)c * np]; if(!TBL) Bn[c] = a[(size_t)(PW + c) * np]; }
        double rec[GEOAC_SEGW], tb[TBL ? 1 : 19];                 // the spline record and the table entry in hand (k, ent: which)
        int k = -1, ent = -1;
        double ref[3] = { 0.15915494309189532, 0.0, 1.0 };                      // Global: reference point of the midpoint latitudes' sin / cos (pp_geom)
        rec[0] = 1.0; rec[1] = 0.0;
        #pragma unroll
        for(int c = 2; c < GEOAC_SEGW; c++) rec[c] = 0.0;
        if(TBL){
            #pragma unroll
            for(int c = 0; c < 19; c++) ltb[c * 256] = 0.0;
        } else {
            #pragma unroll
            for(int c = 0; c < 19; c++) tb[c] = 0.0;
        }
        for(int i = i0; i < i1; i++){
            if(TBL){                                              // (no prefetch: the other three waves of the SIMD cover the wait)
                const double* b = P.path + ((size_t)(i + 1) * PW) * np + col;
                #pragma unroll
                for(int c = 0; c < PW; c++) B[c] = b[(size_t)c * np];
            } else {
                #pragma unroll
                for(int c = 0; c < PW; c++) B[c] = Bn[c];
            }
            if(!TBL && i + 2 <= i1){                              // the row after next, while this segment is evaluated
                const double* b = P.path + ((size_t)(i + 2) * PW) * np + col;
                #pragma unroll
                for(int c = 0; c < PW; c++) Bn[c] = b[(size_t)c * np];
            }
            PPGeom G;
            EQ::pp_geom(P, aux, A, B, G, ref);
            const double xe = clampq(G.x, P.x_min, P.x_max);
            if(!((xe >= rec[0]) & (xe <= rec[1]))){               // (also the first segment: rec[0] > rec[1])
                if constexpr (ONETRIP){   // (GeoacDevParams::pp_onetrip picks the instantiation)
                    // ONE trip to memory instead of five dependent ones (the walk's node reads, the record, the table entry): the neighbour the midpoint
                    // left towards (first segment: the guess by multiplication) is almost always the answer of seg_find, so its record AND its table
                    // entry are fetched together and the walk runs only if the record does not hold xe (then from there: the same k, the same bits).
                    // For the fans that fill the chip (config 3: 375 -> 357 ms per pass).  Hybrid fans keep the slow form below: their post-pass is
                    // hidden either way, and the faster it runs beside the RK4 launches the slower THOSE run (metric fan: post-pass 133 -> 108 ms,
                    // RK4 launch 13.38 -> 13.71 ms, pass 123.4 -> 126.4 ms, A/B in turn on one box)
                    const int kg = k < 0 ? (int)((xe - P.x_min) * P.seg_per_x) : (xe > rec[1] ? k + 1 : k - 1);
                    const int kn = kg < 0 ? 0 : (kg > P.nseg - 1 ? P.nseg - 1 : kg);
                    const int eg = G.x < P.x_min ? P.nseg : (G.x > P.x_max ? P.nseg + 1 : kn);
                    const double* p = P.seg + (size_t)kn * GEOAC_SEGW;
                    const double* q = P.atab + (size_t)eg * GEOAC_ATABW;
                    #pragma unroll
                    for(int c = 0; c < GEOAC_SEGW; c++) rec[c] = p[c];
                    if(TBL){
                        #pragma unroll
                        for(int c = 0; c < 19; c++) ltb[c * 256] = q[c];
                    } else {
                        #pragma unroll
                        for(int c = 0; c < 19; c++) tb[c] = q[c];
                    }
                    ent = eg; k = kn;
                    if(__builtin_expect(!(((xe >= rec[0]) | (kn == 0)) & ((xe <= rec[1]) | (kn == P.nseg - 1))), 0)){
                        k = seg_find(P.seg, P.nseg, xe, kn);
                        const double* p2 = P.seg + (size_t)k * GEOAC_SEGW;
                        #pragma unroll
                        for(int c = 0; c < GEOAC_SEGW; c++) rec[c] = p2[c];
                    }
                } else {
                    k = seg_find(P.seg, P.nseg, xe, k < 0 ? (int)((xe - P.x_min) * P.seg_per_x) : k);
                    const double* p = P.seg + (size_t)k * GEOAC_SEGW;
                    #pragma unroll
                    for(int c = 0; c < GEOAC_SEGW; c++) rec[c] = p[c];
                }
            }
            double t; bool out;
            const int e = atab_locate(P, G.x, xe, k, rec[0], t, out);
            if(e != ent){
                const double* q = P.atab + (size_t)e * GEOAC_ATABW;
                if(TBL){
                    #pragma unroll
                    for(int c = 0; c < 19; c++) ltb[c * 256] = q[c];
                } else {
                    #pragma unroll
                    for(int c = 0; c < 19; c++) tb[c] = q[c];
                }
                ent = e;
            }
            double T, u, v; seg_eval_f(rec, xe, T, u, v);
            const double tt = EQ::pp_tt(P, aux, G, T, u, v);
            bool bad; double at;
            if(TBL){ const double e0 = ltb[0]; bad = out | (e0 < 0.0); at = atab_eval_lds(ltb, e0, t) * G.ds_at; }
            else { bad = out | (tb[0] < 0.0); at = atab_eval(tb, t) * G.ds_at; }
            if(bad){                                              // not served by the table (rare): listed for k_ppfix, which evaluates it exactly
                atomicAdd(&P.counters[GEOAC_CNT_PPFLAG + 1], 1ull);                     // (statistics: geoac_abs_table_info)
                const unsigned long long q = atomicAdd(&P.counters[GEOAC_CNT_PPFLAG], 1ull);
                if(q < (unsigned long long)P.ppfix_cap){ P.ppfix[2 * q] = col; P.ppfix[2 * q + 1] = i; }
                else atomicOr(&P.counters[2], 16ull);             // (list full: the host repeats the fan with the exact post-pass)
            }
            double* o = P.contrib + ((size_t)i * 2) * np + col;
            o[0]  = tt;
            o[np] = at;
            #pragma unroll
            for(int c = 0; c < PW; c++) A[c] = B[c];
        }
    }
}

// k_ppfix: the segments k_postpass_tab listed (the table did not serve them: a flagged entry, a midpoint beyond the strips) - attenuation by
// the exact routine, as k_postpass computes it, with the segment geometry of k_postpass_tab.  One thread per list entry; usually none.
template <class EQ>
__global__ void __launch_bounds__(256) k_ppfix(GeoacDevParams P){
    constexpr int PW = EQ::PW;
    const size_t np = (size_t)P.n_pad;
    unsigned long long n = P.counters[GEOAC_CNT_PPFLAG];
    if(n > (unsigned long long)P.ppfix_cap) n = (unsigned long long)P.ppfix_cap;
    for(unsigned long long q = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (unsigned long long)gridDim.x * blockDim.x){
        const int col = P.ppfix[2 * q], i = P.ppfix[2 * q + 1];
        const int slot = P.colmap ? P.colmap[col] : col;
        double aux[2] = { 0.0, 0.0 };
        EQ::pp_aux(P, P.state + slot, np, aux);
        const double* a = P.path + ((size_t)i * PW) * np + col;
        double A[PW], B[PW];
        #pragma unroll
        for(int c = 0; c < PW; c++){ A[c] = a[(size_t)c * np]; B[c] = a[(size_t)(PW + c) * np]; }
        double ref[3] = { 0.15915494309189532, 0.0, 1.0 };
        PPGeom G;
        EQ::pp_geom(P, aux, A, B, G, ref);
        const double xe = clampq(G.x, P.x_min, P.x_max);
        const int k = seg_find(P.seg, P.nseg, xe, (int)((xe - P.x_min) * P.seg_per_x));
        double T, u, v; seg_eval_f(P.seg + (size_t)k * GEOAC_SEGW, xe, T, u, v);
        const double qT = kGamR * T;
        P.contrib[((size_t)i * 2 + 1) * np + col] = suthbass_alpha(P, G.x - P.r_earth, qT * frsq(qT), rho_eval(P, k, xe), P.freq, P.T_o, P.P_o, P.cbrt_To) * G.ds_at;
    }
}

// k_atab_build: one thread per table entry (see atab_eval).  Entry e < nseg: spline segment e; nseg: the strip [x_min - D, x_min];
// nseg + 1: [x_max, x_max + D].  The exact value at abscissa x is what the exact post-pass computes there: medium at the clamped abscissa,
// height x - r_earth unclamped.
__global__ void __launch_bounds__(64) k_atab_build(GeoacDevParams P, double* __restrict__ tab, double tol){
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if(e >= P.nseg + 2){
        // behind the table: sin / cos of the multiples of 2^-7 rad (GeoacDevParams::lat_trig), by the routine pp_geom would call
        const int i = e - (P.nseg + 2);
        if(i < GEOAC_LAT_N){
            double* o = tab + (size_t)(P.nseg + 2) * GEOAC_ATABW + 2 * (size_t)i;
            fsincos((double)(i - GEOAC_LAT_OFF) * (1.0 / 128.0), o[0], o[1]);
        }
        return;
    }
    double x0, h;
    if(e < P.nseg){ x0 = P.seg[(size_t)e * GEOAC_SEGW]; h = P.seg[(size_t)e * GEOAC_SEGW + 1] - x0; }
    else if(e == P.nseg){ x0 = P.x_min - P.atab_D; h = P.atab_D; }
    else { x0 = P.x_max; h = P.atab_D; }
    const int k = (e < P.nseg) ? e : (e == P.nseg ? 0 : P.nseg - 1);
    const double* p = P.seg + (size_t)k * GEOAC_SEGW;
    auto exact = [&](double s, double* parts) -> double {
        const double x = x0 + 0.5 * (s + 1.0) * h;
        const double xe = clampd(x, P.x_min, P.x_max);
        double T, u, v; seg_eval_f(p, xe, T, u, v);
        const double qT = kGamR * T;
        const double c = qT * frsq(qT);
        const double rho = rho_eval(P, k, xe);
        return parts ? suthbass_alpha<true>(P, x - P.r_earth, c, rho, P.freq, P.T_o, P.P_o, P.cbrt_To, parts)
                     : suthbass_alpha<false>(P, x - P.r_earth, c, rho, P.freq, P.T_o, P.P_o, P.cbrt_To);
    };
    double f[3][6];
    #pragma unroll
    for(int j = 0; j < 6; j++){
        double pr[3];
        exact(cospi((j + 0.5) / 6.0), pr);
        f[0][j] = pr[0]; f[1][j] = pr[1]; f[2][j] = pr[2];
    }
    double* o = tab + (size_t)e * GEOAC_ATABW;
    #pragma unroll
    for(int q = 0; q < 3; q++){
        double cc[6];
        #pragma unroll
        for(int m = 0; m < 6; m++){
            double acc = 0.0;
            #pragma unroll
            for(int j = 0; j < 6; j++) acc += f[q][j] * cospi(m * (j + 0.5) / 6.0);
            cc[m] = acc * (1.0 / 3.0);
        }
        double* w = o + 1 + 6 * q;                                // Chebyshev series -> powers of s
        w[0] = 0.5 * cc[0] - cc[2] + cc[4];
        w[1] = cc[1] - 3.0 * cc[3] + 5.0 * cc[5];
        w[2] = 2.0 * cc[2] - 8.0 * cc[4];
        w[3] = 4.0 * cc[3] - 20.0 * cc[5];
        w[4] = 8.0 * cc[4];
        w[5] = 16.0 * cc[5];
    }
    o[0] = 2.0 / h;
    // the reassembled alpha against the exact routine at points that are not the nodes
    const double chk[8] = { -0.9999, -0.83, -0.5, -0.17, 0.2, 0.55, 0.87, 0.9999 };
    double worst = 0.0;
    bool ok = true;
    for(int q = 0; q < 8; q++){
        const double fx = exact(chk[q], nullptr);
        const double px = atab_eval(o, 0.5 * (chk[q] + 1.0) * h);
        if(!(fx > 0.0) || !(px == px)) ok = false;
        const double err = fabs(px - fx) / fabs(fx);
        if(!(err <= tol)) ok = false;
        worst = (err > worst) ? err : worst;
    }
    if(!ok) o[0] = -o[0];                                       // flagged: not served
    o[GEOAC_ATABW - 1] = worst;
}

// ------------------------------------------------------------------------------------------------
// k_arrival: the arrival part of the records (GeoAcGlobal_main.cpp:296-317 and twins) from the leg's last row, which
// k_rk4 left in the record.  One thread per (ray, leg); runs once, behind the fan's last RK4 launch.
// ------------------------------------------------------------------------------------------------
template <class EQ>
__global__ void __launch_bounds__(256) k_arrival(GeoacDevParams P){
    const int legs = P.bounces + 1;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if(id >= (long long)P.n_rays * legs) return;
    const int slot = (int)(id / legs), leg = (int)(id % legs);
    double* R = P.rec + ((size_t)(P.perm ? P.perm[slot] : slot) * legs + leg) * GEOAC_REC_STRIDE;
    if(R[GEOAC_REC_VALID] == 0.0) return;
    const size_t np = (size_t)P.n_pad;
    const double* st = P.state + slot;
    RayCtx C;
    C.ckey = -1; C.kxy = -1;
    C.c0 = st[ST_C0 * np]; C.nu0 = st[ST_NU0 * np];
    #pragma unroll
    for(int q = 0; q < 6; q++) C.a[q] = st[(ST_AUX0 + q) * np];     // (3-D / 2-D: the constants of the ray; Global: not read)
    double yl[18];
    #pragma unroll
    for(int e = 0; e < 18; e++) yl[e] = R[GEOAC_REC_STATE + e];
    EQ::arrival(P, C, slot, yl, R);
}

// ------------------------------------------------------------------------------------------------
// k_accum: in-order summation per ray, leg bookkeeping (GeoAcGlobal_main.cpp:256-291, Q7)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_accum(GeoacDevParams P){
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if(col >= (P.colmap ? *P.n_cols : P.n_pad)) return;
    const size_t np = (size_t)P.n_pad;
    const int nr = P.nrows[col];                                // 0 for slots without a ray
    if(nr < 2) return;
    const int slot = P.colmap ? P.colmap[col] : col;
    double* st = P.state + slot;
    double tt = st[ST_TT * np], at = st[ST_AT * np];           // cumulative over the ray
    double ltt = st[ST_LTT * np], lat = st[ST_LAT * np];   // per-leg partial sums (arrivals-only form)
    int leg = (int)st[ST_PLEG * np];
    const int ne = P.nlegend[col];
    int e = 0;
    int next_end = (e < ne) ? P.legend[(size_t)e * np + col] : 0x7fffffff;
    int cur_end = -1;
    const bool rays_form = P.rays_form != 0;      // WriteRays/WriteCaustics form of Q7; GeoAc2D always uses it (GeoAc2D_main.cpp:190-192)
    const int nev = P.ev_cap > 0 ? P.nev[col] : 0;
    int ev = 0;
    auto emit_events = [&](int crow){
        while(ev < nev && P.ev_row[(size_t)ev * np + col] == crow){
        const int mm = P.ev_m[(size_t)ev * np + col];
        const int kind = (mm >> 30) & 1, m = mm & 0x3fffffff;
        unsigned long long o = atomicAdd(&P.counters[3], 1ull);
        if(o < (unsigned long long)P.smp_cap){
            double* S = P.smp_out + o * GEOAC_SMP_STRIDE;
            const double* row = P.path + ((size_t)crow * P.pathw) * np + col;
            S[GEOAC_SMP_RAY] = (double)(P.perm ? P.perm[slot] : slot); S[GEOAC_SMP_LEG] = (double)leg; S[GEOAC_SMP_M] = (double)m; S[GEOAC_SMP_KIND] = (double)kind;
            double amp = P.ev_amp[(size_t)ev * np + col];
            double amp_db = P.calc_amp ? 20.0 * log10(amp) : 0.0;
            // position columns, then (raypath row) amplitude [dB], -attenuation, travel time or (caustic row) travel time - written out by case: a
            // register array indexed by a run-time column count would live in scratch memory
            double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, v5 = 0.0;
            const bool sph = (P.eqset == GEOAC_EQ_GLOBAL || P.eqset == GEOAC_EQ_GLOBAL_RNGDEP);
            if(P.eqset == GEOAC_EQ_2D){
                v0 = row[0]; const double z = row[np]; v1 = (kind == 0) ? (z < 0.0 ? 0.0 : z) : z;
                if(kind == 0){ v2 = amp_db; v3 = -at; v4 = tt; } else v2 = tt;
            } else {
                if(sph){ v0 = row[0] - P.r_earth; v1 = row[np] * 180.0 / kPi; v2 = row[2 * np] * 180.0 / kPi; }
                else { v0 = row[0]; v1 = row[np]; const double z = row[2 * np];
                       v2 = (P.eqset == GEOAC_EQ_3D || kind == 0) ? (z < 0.0 ? 0.0 : z) : z; }       // max(z, 0) in both files of GeoAc3D (GeoAc3D_main.cpp:257,267); RngDep caustic rows: raw z
                if(kind == 0){ v3 = amp_db; v4 = -at; v5 = tt; }
                else if(P.eqset == GEOAC_EQ_3D_RNGDEP){ v3 = 0.0; v4 = tt; }                         // caustic rows of GeoAc3D.RngDep: a 0.0 column (GeoAc3D.RngDep_main.cpp:279-284)
                else v3 = tt;
            }
            S[GEOAC_SMP_V0 + 0] = v0; S[GEOAC_SMP_V0 + 1] = v1; S[GEOAC_SMP_V0 + 2] = v2; S[GEOAC_SMP_V0 + 3] = v3; S[GEOAC_SMP_V0 + 4] = v4; S[GEOAC_SMP_V0 + 5] = v5;
        }
        ev++;
    }
    };
    emit_events(0);             // events on the carry row (the row a previous epoch ended on)
    // (measured, round 3: a separate summation loop for arrivals-only fans with eight rows' loads in flight - config 3 373 -> 388 ms per pass: the sums
    //  run beside the next epoch's RK4 and post-pass, and pulling their rows faster only takes bandwidth from those)
    int i_begin = 0;
    if(P.accum_batch && !rays_form && nev == 0){
        // Late epochs of an arrivals-only fan (a few rays still alive: nothing else for the loads to compete with, and these sums are the uncovered tail of the fan): the
        // contributions of eight rows are fetched together and added one after the other in the same order - the plain loop below waits out a trip to memory per row
        // (~0.7 us x 2 048 rows = 1.5 ms behind the last RK4 epoch of the metric fan).  The same sums, the same bits.
        for(int i0 = 0; i0 + 1 < nr; i0 += 8){
            double c0[8], c1[8];
            #pragma unroll
            for(int j = 0; j < 8; j++){
                const int i = i0 + j;
                const double* cpt = P.contrib + ((size_t)(i + 1 < nr ? i : 0) * 2) * np + col;
                c0[j] = cpt[0]; c1[j] = cpt[np];
            }
            #pragma unroll
            for(int j = 0; j < 8; j++){
                const int i = i0 + j;
                if(i + 1 >= nr || i == cur_end) continue;
                ltt += c0[j]; lat += c1[j];
                if(i + 1 == next_end){
                    tt += ltt; at += lat; ltt = 0.0; lat = 0.0;
                    double* R = P.rec + ((size_t)(P.perm ? P.perm[slot] : slot) * (P.bounces + 1) + leg) * GEOAC_REC_STRIDE;
                    R[GEOAC_REC_TTIME] = tt;
                    R[GEOAC_REC_ATTEN] = at;
                    leg++; cur_end = i + 1; e++;
                    next_end = (e < ne) ? P.legend[(size_t)e * np + col] : 0x7fffffff;
                }
            }
        }
        i_begin = nr;
    }
    for(int i = i_begin; i + 1 < nr; i++){
        if(i == cur_end) continue;                              // (leg-end row -> next leg's start row): not a segment
        const double* cpt = P.contrib + ((size_t)i * 2) * np + col;
        const bool last = (i + 1 == next_end);
        if(rays_form){
            if(!last){ tt += cpt[0]; at += cpt[np]; }           // segments 0..k-2 only (GeoAcGlobal_main.cpp:264-267)
        } else {
            ltt += cpt[0]; lat += cpt[np];                      // GeoAc_TravelTime / GeoAc_SB_Atten start from 0 per leg
        }
        emit_events(i + 1);     // raypath / caustic rows at chunk row i+1: the sums now include segment (m-1, m) (GeoAcGlobal_main.cpp:266-284)
        if(last){
            if(!rays_form){ tt += ltt; at += lat; ltt = 0.0; lat = 0.0; }
            double* R = P.rec + ((size_t)(P.perm ? P.perm[slot] : slot) * (P.bounces + 1) + leg) * GEOAC_REC_STRIDE;
            R[GEOAC_REC_TTIME] = tt;
            R[GEOAC_REC_ATTEN] = at;
            leg++; cur_end = i + 1; e++;
            next_end = (e < ne) ? P.legend[(size_t)e * np + col] : 0x7fffffff;
        }
    }
    st[ST_TT * np] = tt; st[ST_AT * np] = at; st[ST_PLEG * np] = (double)leg;
    st[ST_LTT * np] = ltt; st[ST_LAT * np] = lat;
}

// ------------------------------------------------------------------------------------------------
// k_compact: the dense, order-preserving list of the rays still alive after an epoch (north_star: "wavefront ballot for ground-bounce /
// termination compaction").  One workgroup walks the previous list in tiles of 1024 columns: a wave ballot of the live flags, the
// popcount below each lane and a running offset give every live ray its column of the next epoch.  The order of the list (launch
// inclination) is kept, so the next epoch's waves are again made of neighbouring rays.  n <= ~1e6 columns: 10-200 us per epoch.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_compact(GeoacDevParams P, const int* __restrict__ cur, const int* __restrict__ n_cur, int n_first,
                                                  int* __restrict__ next, int* __restrict__ n_next){
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int n = cur ? *n_cur : n_first;                       // cur == NULL: the first epoch ran over the identity list 0..n_first-1
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t np = (size_t)P.n_pad;
    if(threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for(int t0 = 0; t0 < n; t0 += 1024){
        const int c = t0 + threadIdx.x;
        const int slot = (c < n) ? (cur ? cur[c] : c) : -1;
        const bool live = slot >= 0 && P.state[ST_DONE * np + slot] == 0.0;
        const unsigned long long m = __ballot(live);
        if(lane == 0) wave_cnt[wv] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for(int w = 0; w < wv; w++) off += wave_cnt[w];
        if(live) next[off + __popcll(m & ((1ull << lane) - 1ull))] = slot;
        __syncthreads();
        if(threadIdx.x == 0){ int tot = 0; for(int w = 0; w < 16; w++) tot += wave_cnt[w]; base_s += tot; }
        __syncthreads();
    }
    if(threadIdx.x == 0) *n_next = base_s;
}

// ------------------------------------------------------------------------------------------------
// device-function probes (include/geoac_probe.h): one thread per point through the very functions the RK4 and post-pass kernels call,
// so that the table lookups and the absorption model have parity tests of their own (not only through fan integrals)
// ------------------------------------------------------------------------------------------------
// c, c', c'', u, u', u'', v, v', v'' and rho of the 1-D atmosphere at abscissa x: seg_guess + seg_eval (the RHS's lookup) and the sound-speed
// algebra of global_rhs / cart3_rhs (rsq-based c, c' = gamR/(2c) T', c'' = gamR/(2c) T'' - c'^2/c)
__global__ void __launch_bounds__(256) k_probe_atmo1d(GeoacDevParams P, int n, const double* __restrict__ x, double* __restrict__ out9, double* __restrict__ rho){
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    const double xe = clampd(x[i], P.x_min, P.x_max);
    const int k = seg_guess(P.seg, P, xe);
    Atm9 a; seg_eval<true>(P.seg, k, xe, a);
    const double qT = kGamR * a.T;
    const double ic = frsq(qT);
    const double c = qT * ic;
    const double hc = (0.5 * kGamR) * ic;
    const double dc = hc * a.dT;
    const double ddc = __builtin_fma(hc, a.ddT, -(dc * dc) * ic);
    double* o = out9 + (size_t)9 * i;
    o[0] = c; o[1] = dc; o[2] = ddc; o[3] = a.u; o[4] = a.du; o[5] = a.ddu; o[6] = a.v; o[7] = a.dv; o[8] = a.ddv;
    rho[i] = rho_eval(P, k, xe);
}

// SuthBass_Alpha at abscissa x[i] (geocentric radius for the spherical sets) and frequency f[i]: medium_at + suthbass_alpha as the
// post-pass calls them (1-D sets; reference state T_o, P_o from the launch parameters)
__global__ void __launch_bounds__(256) k_probe_absorption(GeoacDevParams P, int n, const double* __restrict__ x, const double* __restrict__ f, double* __restrict__ out){
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    const Medium m = medium_at(P, x[i]);
    out[i] = suthbass_alpha(P, x[i] - P.r_earth, m.c, m.rho, f[i], P.T_o, P.P_o, P.cbrt_To);
}

// the absorption table at abscissa x[i] (atab_alpha as k_postpass_tab calls it; -1 where the table does not serve the point)
__global__ void __launch_bounds__(256) k_probe_atab(GeoacDevParams P, int n, const double* __restrict__ x, double* __restrict__ out){
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    const double xe = clampd(x[i], P.x_min, P.x_max);
    const int k = seg_guess_mul(P.seg, P, xe);
    out[i] = atab_alpha(P, x[i], xe, k, P.seg[(size_t)k * GEOAC_SEGW]);
}

// grid interpolant at (a0, a1, a2) in table order ((x, y, z) Cartesian, (lat, lon, r) spherical): out30 = the ten outputs of
// grid_eval_all / grid_eval3_coop for T, u, v (table order of geoac_rngdep.h); api7 = c, rho, u, v, dc/dz, du/dz, dv/dz of medium3_at.
// coop != 0: whole waves through the cooperative gather (blocks of 64, LDS exchange buffer; n is padded by repeating the last point)
template <bool GLB>
__global__ void __launch_bounds__(64) k_probe_grid(GeoacDevParams P, int n, const double* __restrict__ a0, const double* __restrict__ a1, const double* __restrict__ a2,
                                                   int coop, double* __restrict__ out30, double* __restrict__ api7){
    extern __shared__ double lds_tab[];
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = i0 < n ? i0 : n - 1;
    const double x = clampd(a0[i], P.gx[0], P.gx[P.gnx - 1]), y = clampd(a1[i], P.gy[0], P.gy[P.gny - 1]), z = clampd(a2[i], P.x_min, P.x_max);
    GridLoc L; grid_locate(P, x, y, z, -1, L);
    double M[3][10];
    if(coop){
        if constexpr ((GRec<GLB>::PACKED || GLB) && GEOAC_COOP_GLDS) grid_eval3_glds<true, GLB>(P, L, M, (char*)lds_tab);
        else if constexpr (GRec<GLB>::PACKED) grid_eval3_coop8<true>(P, L, M, (char*)lds_tab);
        else grid_eval3_coop<true, GLB>(P, L, M, (char*)lds_tab);
    }
    else {
        grid_eval_all<true, GLB, 1>(P, 0, L, M[0], 0);
        grid_eval_all<true, GLB, 1>(P, 1, L, M[1], 0);
        grid_eval_all<true, GLB, 1>(P, 2, L, M[2], 0);
    }
    if(i0 >= n) return;
    #pragma unroll
    for(int f = 0; f < 3; f++){
        #pragma unroll
        for(int q = 0; q < 10; q++) out30[(size_t)30 * i + 10 * f + q] = M[f][q];
    }
    const Medium3 m = medium3_at<true, true, GLB>(P, a0[i], a1[i], a2[i]);
    double* a = api7 + (size_t)7 * i;
    a[0] = m.c; a[1] = m.rho; a[2] = m.u; a[3] = m.v; a[4] = m.dcz; a[5] = m.duz; a[6] = m.dvz;
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers (called from geoac_api.cpp)
// ------------------------------------------------------------------------------------------------
#ifndef GEOAC_NO_LAUNCHERS          // tools/dev_kernel.sh compiles single instantiations of the kernels above
#define GEOAC_DISPATCH_EQ(P, CALL) \
    switch((P)->eqset * 2 + ((P)->calc_amp ? 1 : 0)){ \
        case GEOAC_EQ_GLOBAL * 2 + 1: { using EQ = EqGlobal<true>;  CALL; } break; \
        case GEOAC_EQ_GLOBAL * 2 + 0: { using EQ = EqGlobal<false>; CALL; } break; \
        case GEOAC_EQ_3D * 2 + 1:     { using EQ = Eq3D<true>;      CALL; } break; \
        case GEOAC_EQ_3D * 2 + 0:     { using EQ = Eq3D<false>;     CALL; } break; \
        case GEOAC_EQ_2D * 2 + 1:     { using EQ = Eq2D<true>;      CALL; } break; \
        case GEOAC_EQ_2D * 2 + 0:     { using EQ = Eq2D<false>;     CALL; } break; \
        case GEOAC_EQ_3D_RNGDEP * 2 + 1: { using EQ = Eq3DRngDep<true>;  CALL; } break; \
        case GEOAC_EQ_3D_RNGDEP * 2 + 0: { using EQ = Eq3DRngDep<false>; CALL; } break; \
        case GEOAC_EQ_GLOBAL_RNGDEP * 2 + 1: { using EQ = EqGlobalRngDep<true>;  CALL; } break; \
        case GEOAC_EQ_GLOBAL_RNGDEP * 2 + 0: { using EQ = EqGlobalRngDep<false>; CALL; } break; \
        default: return hipErrorNotSupported; }

// the grid sets' four-lane kernels without the record cache (fans of 4 097 - 16 384 rays with amplitudes, whose records do not fit the LDS: in the launch
// plan) and, in A/B builds only, their two-lane kernels (GRID_LANES=2)
#define GEOAC_DISPATCH_GRID4(P, CALL) \
    else if((P)->lanes_per_ray == 4){ \
        switch((P)->eqset * 2 + ((P)->calc_amp ? 1 : 0)){ \
            case GEOAC_EQ_3D_RNGDEP * 2 + 1: { using EQ = Eq3DRngDep<true, 4>;  CALL; } break; \
            case GEOAC_EQ_3D_RNGDEP * 2 + 0: { using EQ = Eq3DRngDep<false, 4>; CALL; } break; \
            case GEOAC_EQ_GLOBAL_RNGDEP * 2 + 1: { using EQ = EqGlobalRngDep<true, 4>;  CALL; } break; \
            case GEOAC_EQ_GLOBAL_RNGDEP * 2 + 0: { using EQ = EqGlobalRngDep<false, 4>; CALL; } break; \
            default: return hipErrorNotSupported; } \
    }
#if GEOAC_AB
#define GEOAC_DISPATCH_AB_GRID(P, CALL) GEOAC_DISPATCH_GRID4(P, CALL) \
    else if((P)->lanes_per_ray == 2 && (P)->gtab){ \
        switch((P)->eqset * 2 + ((P)->calc_amp ? 1 : 0)){ \
            case GEOAC_EQ_3D_RNGDEP * 2 + 1: { using EQ = Eq3DRngDep<true, 2>;  CALL; } break; \
            case GEOAC_EQ_3D_RNGDEP * 2 + 0: { using EQ = Eq3DRngDep<false, 2>; CALL; } break; \
            case GEOAC_EQ_GLOBAL_RNGDEP * 2 + 1: { using EQ = EqGlobalRngDep<true, 2>;  CALL; } break; \
            case GEOAC_EQ_GLOBAL_RNGDEP * 2 + 0: { using EQ = EqGlobalRngDep<false, 2>; CALL; } break; \
            default: return hipErrorNotSupported; } \
    }
#else
#define GEOAC_DISPATCH_AB_GRID(P, CALL) GEOAC_DISPATCH_GRID4(P, CALL) \
    else if((P)->lanes_per_ray == 2 && (P)->gtab){ return hipErrorNotSupported; }
#endif

// RK4 only: the grid sets have four-lanes-per-ray variants (small fans)
#define GEOAC_DISPATCH_EQ_RK4(P, CALL) \
    if((P)->lanes_per_ray == 16){ \
        if((P)->eqset == GEOAC_EQ_GLOBAL_RNGDEP && (P)->calc_amp && (P)->quad_cache){ using EQ = EqGlobalRngDepHex; CALL; } \
        else if((P)->eqset == GEOAC_EQ_GLOBAL_RNGDEP && !(P)->calc_amp && (P)->quad_cache){ using EQ = EqGlobalRngDepScan16; CALL; } \
        else return hipErrorNotSupported; \
    } else if((P)->lanes_per_ray == 8){ \
        if((P)->eqset == GEOAC_EQ_GLOBAL_RNGDEP && (P)->calc_amp && (P)->quad_cache){ using EQ = EqGlobalRngDepOct; CALL; } \
        else if((P)->eqset == GEOAC_EQ_3D_RNGDEP && (P)->calc_amp && (P)->quad_cache){ using EQ = Eq3DRngDepOct; CALL; } \
        else return hipErrorNotSupported; \
    } else if((P)->lanes_per_ray == 4 && (P)->quad_cache){ \
        switch((P)->eqset * 2 + ((P)->calc_amp ? 1 : 0)){ \
            case GEOAC_EQ_3D_RNGDEP * 2 + 1: { using EQ = Eq3DRngDep<true, 4, false, true>;  CALL; } break; \
            case GEOAC_EQ_3D_RNGDEP * 2 + 0: { using EQ = Eq3DRngDep<false, 4, false, true>; CALL; } break; \
            case GEOAC_EQ_GLOBAL_RNGDEP * 2 + 1: { using EQ = EqGlobalRngDep<true, 4, false, true>;  CALL; } break; \
            case GEOAC_EQ_GLOBAL_RNGDEP * 2 + 0: { using EQ = EqGlobalRngDep<false, 4, false, true>; CALL; } break; \
            default: return hipErrorNotSupported; } \
    } GEOAC_DISPATCH_AB_GRID(P, CALL) \
    else if((P)->coop && (P)->gtab && (P)->lanes_per_ray == 1 && (P)->spread <= 1){ \
        switch((P)->eqset * 2 + ((P)->calc_amp ? 1 : 0)){ \
            case GEOAC_EQ_3D_RNGDEP * 2 + 1: { using EQ = Eq3DRngDep<true, 1, true>;  CALL; } break; \
            case GEOAC_EQ_3D_RNGDEP * 2 + 0: { using EQ = Eq3DRngDep<false, 1, true>; CALL; } break; \
            case GEOAC_EQ_GLOBAL_RNGDEP * 2 + 1: { using EQ = EqGlobalRngDep<true, 1, true>;  CALL; } break; \
            case GEOAC_EQ_GLOBAL_RNGDEP * 2 + 0: { using EQ = EqGlobalRngDep<false, 1, true>; CALL; } break; \
            default: return hipErrorNotSupported; } \
    } else GEOAC_DISPATCH_EQ(P, CALL)

extern "C" hipError_t geoac_launch_init(const GeoacDevParams* P, hipStream_t s){
    dim3 b(256), g((P->n_pad + 255) / 256);
    GEOAC_DISPATCH_EQ(P, hipLaunchKernelGGL(k_init<EQ>, g, b, 0, s, *P));
    return hipGetLastError();
}

template <class EQ>
static hipError_t launch_rk4_t(const GeoacDevParams* P, int block, hipStream_t s, unsigned* n_wg){
    if(block < 64 || block > 256 || (block % 64) != 0) return hipErrorInvalidValue;   // k_rk4 carries __launch_bounds__(256)
    if(P->slot_lo < 0 || P->slot_hi > P->n_pad || P->slot_lo >= P->slot_hi) return hipErrorInvalidValue;
    const long long lanes = (long long)(P->slot_hi - P->slot_lo) * EQ::LANES * (P->spread > 1 ? P->spread : 1);
    dim3 b(block), g((unsigned)((lanes + block - 1) / block));
    if(P->sub > 1){
        // sub-epochs: sub x sub_w workgroups (k_rk4); the flags are zeroed on the launch stream
        if(!EQ::COOP || block != 64 || !P->sub_flags || P->sub_w < (int)g.x || (P->sub_w & 7)) return hipErrorInvalidValue;
        hipError_t e = hipMemsetAsync(P->sub_flags, 0, sizeof(int) * (size_t)P->sub_w, s);
        if(e != hipSuccess) return e;
        g.x = (unsigned)P->sub_w * (unsigned)P->sub;
    }
    if(n_wg) *n_wg = g.x;
    size_t lds = P->table_in_lds ? (size_t)P->nseg * GEOAC_SEGW * sizeof(double) : 0;
    if(EQ::LDS_STATE) lds = (size_t)(block / 64) * ((EQ::COOP ? EQ::XCHG_BYTES : 0) + 2 * EQ::E * 64 * sizeof(double));   // per wave: exchange buffer, y and yn rows (k_rk4)
    if(EQ::CACHE){
        if(block != 64) return hipErrorInvalidValue;
        lds += GEOAC_CACHE_BYTES + (size_t)(P->nseg + 1) * sizeof(double);                 // per-lane records, z nodes
    }
    const bool smp = (P->mode & (GEOAC_MODE_WRITE_RAYS | GEOAC_MODE_WRITE_CAUSTICS)) != 0;
    if(smp && EQ::SPLIT && !EQ::ROW_SPLIT) return hipErrorNotSupported;                  // (the eight-lane kernel has no sample capture)
    #define GEOAC_RK4_LAUNCH(LDSF, SMPF) do { \
        if(lds > 65536){ \
            hipError_t err = hipFuncSetAttribute((const void*)k_rk4<EQ, LDSF, SMPF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if(err != hipSuccess) return err; \
        } \
        hipLaunchKernelGGL((k_rk4<EQ, LDSF, SMPF>), g, b, lds, s, *P); } while(0)
    if(P->table_in_lds){ if(smp) GEOAC_RK4_LAUNCH(true, true); else GEOAC_RK4_LAUNCH(true, false); }
    else               { if(smp) GEOAC_RK4_LAUNCH(false, true); else GEOAC_RK4_LAUNCH(false, false); }
    #undef GEOAC_RK4_LAUNCH
    return hipGetLastError();
}

// A/B builds (`make AB=1`, -DGEOAC_AB=1) also hold the kernels that were built, measured slower and kept only for comparisons and the schedule-independence
// tests: the wave-specialised k_rk4_duo and the grid sets' two-lane kernels.  The shipped library holds the kernels the launch plan
// can select (and the exact post-pass, its fallback when a table does not serve a profile); geoac_build_has_ab() tells which build this is.
extern "C" int geoac_build_has_ab(void){ return GEOAC_AB ? 1 : 0; }
#if GEOAC_AB
// the wave-specialised kernel of the stratified Global set with amplitudes (geoac_duo.h): 128 rays per workgroup
extern "C" size_t geoac_duo_lds(int nseg){ return geoac_duo_lds_bytes(nseg); }
static hipError_t launch_rk4_duo(const GeoacDevParams* P, hipStream_t s, unsigned* n_wg){
    if(P->eqset != GEOAC_EQ_GLOBAL || !P->calc_amp || P->gtab || (P->mode & (GEOAC_MODE_WRITE_RAYS | GEOAC_MODE_WRITE_CAUSTICS))) return hipErrorNotSupported;
    if(P->slot_lo < 0 || P->slot_hi > P->n_pad || P->slot_lo >= P->slot_hi) return hipErrorInvalidValue;
    const size_t lds = geoac_duo_lds_bytes(P->nseg);
    if(lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 b(256), g((unsigned)((P->slot_hi - P->slot_lo + 127) / 128));
    if(n_wg) *n_wg = g.x;
    #define GEOAC_DUO_LAUNCH(VV) do { \
        hipError_t err = hipFuncSetAttribute((const void*)k_rk4_duo<VV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if(err != hipSuccess) return err; \
        hipLaunchKernelGGL(k_rk4_duo<VV>, g, b, lds, s, *P); } while(0)
    switch(P->duo){
        case 32: GEOAC_DUO_LAUNCH(32); break;    // (timing diagnostics, tools/perf_duo.py: the base wave alone / messages consumed but not used - records are NOT valid)
        case 66: GEOAC_DUO_LAUNCH(66); break;
        case 160: GEOAC_DUO_LAUNCH(160); break;
        default: GEOAC_DUO_LAUNCH(2); break;
    }
    #undef GEOAC_DUO_LAUNCH
    return hipGetLastError();
}

// the wave-specialised kernel of the stratified Global set with amplitudes (geoac_trio.h): 64 rays per workgroup of three waves
extern "C" size_t geoac_trio_lds(int nseg){ return geoac_trio_lds_bytes(nseg); }
static hipError_t launch_rk4_trio(const GeoacDevParams* P, hipStream_t s, unsigned* n_wg){
    if(P->eqset != GEOAC_EQ_GLOBAL || !P->calc_amp || P->gtab || (P->mode & (GEOAC_MODE_WRITE_RAYS | GEOAC_MODE_WRITE_CAUSTICS))) return hipErrorNotSupported;
    if(P->slot_lo < 0 || P->slot_hi > P->n_pad || P->slot_lo >= P->slot_hi) return hipErrorInvalidValue;
    const size_t lds = geoac_trio_lds_bytes(P->nseg);
    if(lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 b(192), g((unsigned)((P->slot_hi - P->slot_lo + 63) / 64));
    if(n_wg) *n_wg = g.x;
    #define GEOAC_TRIO_LAUNCH(VV) do { \
        hipError_t err = hipFuncSetAttribute((const void*)k_rk4_trio<VV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if(err != hipSuccess) return err; \
        hipLaunchKernelGGL(k_rk4_trio<VV>, g, b, lds, s, *P); } while(0)
    switch(P->trio){
        case 3: GEOAC_TRIO_LAUNCH(3); break;     // (timing diagnostics, tools/perf_trio.py: records are NOT valid)
        case 5: GEOAC_TRIO_LAUNCH(5); break;
        case 11: GEOAC_TRIO_LAUNCH(11); break;
        case 27: GEOAC_TRIO_LAUNCH(27); break;
        default: GEOAC_TRIO_LAUNCH(1); break;
    }
    #undef GEOAC_TRIO_LAUNCH
    return hipGetLastError();
}
#else
extern "C" size_t geoac_duo_lds(int nseg){ (void)nseg; return (size_t)1 << 40; }          // (never fits: the plan cannot pick the kernel)
extern "C" size_t geoac_trio_lds(int nseg){ (void)nseg; return (size_t)1 << 40; }
static hipError_t launch_rk4_duo(const GeoacDevParams*, hipStream_t, unsigned*){ return hipErrorNotSupported; }
static hipError_t launch_rk4_trio(const GeoacDevParams*, hipStream_t, unsigned*){ return hipErrorNotSupported; }
#endif
extern "C" hipError_t geoac_launch_rk4(const GeoacDevParams* P, int block, hipStream_t s, unsigned* n_wg){
    if(P->duo) return launch_rk4_duo(P, s, n_wg);
    if(P->trio && P->lanes_per_ray == 2 && !P->gtab && P->eqset == GEOAC_EQ_GLOBAL) return launch_rk4_trio(P, s, n_wg);
    if(P->lanes_per_ray == 2 && !P->gtab && P->eqset == GEOAC_EQ_GLOBAL) return launch_rk4_t<EqGlobalPair>(P, block, s, n_wg);
    if(P->lanes_per_ray == 2 && !P->gtab && P->eqset == GEOAC_EQ_3D) return launch_rk4_t<Eq3DPair>(P, block, s, n_wg);
    GEOAC_DISPATCH_EQ_RK4(P, return launch_rk4_t<EQ>(P, block, s, n_wg));
    return hipErrorNotSupported;
}

// k_gate: holds a stream until `expected` RK4 workgroups (counted from the start of the fan) have become resident.  The post-pass of
// epoch e-1 and the RK4 launch of epoch e become runnable at the same moment; the post-pass has thousands of short workgroups that
// refill every CU as fast as they drain, and an RK4 workgroup (a whole CU: 153 KB of LDS, 4 x 384 registers) is then not placed before
// the post-pass grid is exhausted - the epoch degenerates to post-pass followed by RK4.  With the gate the RK4 workgroups take their CUs
// first and the post-pass fills what is left.  Bounded spin (~40 ms): the gate always exits, a late RK4 launch only costs the ordering.
__global__ void __launch_bounds__(64) k_gate(const unsigned long long* counter, unsigned long long expected){
    if(threadIdx.x != 0) return;
    for(int it = 0; it < 20000; it++){
        if(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= expected) break;
        __builtin_amdgcn_s_sleep(64);
    }
}

extern "C" hipError_t geoac_launch_gate(const GeoacDevParams* P, unsigned long long expected, hipStream_t s){
    hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, s, (const unsigned long long*)(P->counters + 5), expected);
    return hipGetLastError();
}

extern "C" hipError_t geoac_launch_postpass(const GeoacDevParams* P, int rows, hipStream_t s){
    if(rows < 2) return hipSuccess;
    long long total = (P->gtab && GEOAC_PP_TILE) ? (long long)((P->n_cols_bound + 15) / 16) * ((rows - 1 + 15) / 16)       // grid sets: 16 x 16 tiles (k_postpass)
                              : (long long)((P->n_cols_bound + 255) / 256) * (rows - 1);
    long long nbl = P->pp_blocks > 0 ? P->pp_blocks : total;
    if(total < nbl) nbl = total;
    if(nbl > 0x7fffffffLL) nbl = 0x7fffffffLL;
    int nb = (int)nbl;
    dim3 b(256), g(nb);
    GEOAC_DISPATCH_EQ(P, hipLaunchKernelGGL(k_postpass<EQ>, g, b, EQ::PP_DEDUP ? 4 * GEOAC_PP_SLOTS * GEOAC_PP_SLOTB : 0, s, *P, rows));
    return hipGetLastError();
}

// post-pass of a stratified set through the absorption table (k_postpass_tab; the few segments the table does not serve are listed and
// evaluated exactly by k_ppfix behind it)
extern "C" hipError_t geoac_launch_postpass_tab(const GeoacDevParams* P, int rows, hipStream_t s){
    if(rows < 2) return hipSuccess;
    if(!P->atab || P->gtab) return hipErrorInvalidValue;
    const int gy = (rows - 1 + GEOAC_PP_ROWS - 1) / GEOAC_PP_ROWS;           // groups of GEOAC_PP_ROWS rows: the grid's y (at most 65 535 per launch)
    dim3 b(256);
    // LDS the kernel never touches (GeoacDevParams::pp_lds_pad, set by the launch plan): a workgroup that asks for more than 7 KiB cannot be placed
    // on a CU that holds an RK4 workgroup with its table (153 KiB of the 160), one that asks for more than 80 KiB is alone on its CU (one wave per
    // SIMD).  With the exact fall-back out of the kernel its waves are small enough to sit beside an RK4 wave on the same SIMD and to run at three
    // or four waves per SIMD: right for a fan whose RK4 launches fill the chip (config 3: +11 %), wrong for the fans whose time is the serial chain of
    // one ray (GeoAc3D 360 x 90: +6 % per pass beside the RK4 waves, +2.5 % at full occupancy on the free CUs, 0 at one wave per SIMD there)
    unsigned pad = (unsigned)P->pp_lds_pad;
    int v = (P->eqset * 2 + (P->calc_amp ? 1 : 0)) * 2 + (P->pp_onetrip ? 1 : 0);
    void (*f)(GeoacDevParams, int, int) = nullptr;
    if(P->pp_lds_table && P->eqset == GEOAC_EQ_GLOBAL && P->pp_onetrip){
        // the table entry in LDS (19 x 256 doubles per workgroup), four waves per SIMD: the fans of the spherical set that fill the chip
        f = P->calc_amp ? k_postpass_tab<EqGlobal<true>, true, true> : k_postpass_tab<EqGlobal<false>, true, true>;
        if(pad < 19u * 256u * 8u) pad = 19u * 256u * 8u;
        v = 24 + (P->calc_amp ? 1 : 0);
    } else
    switch(v){
        case (GEOAC_EQ_GLOBAL * 2 + 1) * 2 + 0: f = k_postpass_tab<EqGlobal<true>, false>; break;
        case (GEOAC_EQ_GLOBAL * 2 + 1) * 2 + 1: f = k_postpass_tab<EqGlobal<true>, true>; break;
        case (GEOAC_EQ_GLOBAL * 2 + 0) * 2 + 0: f = k_postpass_tab<EqGlobal<false>, false>; break;
        case (GEOAC_EQ_GLOBAL * 2 + 0) * 2 + 1: f = k_postpass_tab<EqGlobal<false>, true>; break;
        case (GEOAC_EQ_3D * 2 + 1) * 2 + 0:     f = k_postpass_tab<Eq3D<true>, false>; break;
        case (GEOAC_EQ_3D * 2 + 1) * 2 + 1:     f = k_postpass_tab<Eq3D<true>, true>; break;
        case (GEOAC_EQ_3D * 2 + 0) * 2 + 0:     f = k_postpass_tab<Eq3D<false>, false>; break;
        case (GEOAC_EQ_3D * 2 + 0) * 2 + 1:     f = k_postpass_tab<Eq3D<false>, true>; break;
        case (GEOAC_EQ_2D * 2 + 1) * 2 + 0:     f = k_postpass_tab<Eq2D<true>, false>; break;
        case (GEOAC_EQ_2D * 2 + 1) * 2 + 1:     f = k_postpass_tab<Eq2D<true>, true>; break;
        case (GEOAC_EQ_2D * 2 + 0) * 2 + 0:     f = k_postpass_tab<Eq2D<false>, false>; break;
        case (GEOAC_EQ_2D * 2 + 0) * 2 + 1:     f = k_postpass_tab<Eq2D<false>, true>; break;
        default: return hipErrorNotSupported;
    }
    if(pad > 65536u){
        static bool raised[64][32] = {};                             // (a function attribute is per device)
        int dev = 0;
        if(hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
        if(v >= 0 && v < 32 && (dev == 63 || !raised[dev][v])){
            hipError_t err = hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if(err != hipSuccess) return err;
            raised[dev][v] = true;
        }
    }
    for(int gy0 = 0; gy0 < gy; gy0 += 65535){
        dim3 g((unsigned)((P->n_cols_bound + 255) / 256), (unsigned)(gy - gy0 > 65535 ? 65535 : gy - gy0));
        hipLaunchKernelGGL(f, g, b, pad, s, *P, rows, gy0);
    }
    hipError_t e = hipGetLastError();
    if(e != hipSuccess) return e;
    // the listed segments, exactly (the list's counter was zeroed on this stream before the launch above: geoac_api.cpp)
    dim3 gf(64);
    switch(P->eqset * 2 + (P->calc_amp ? 1 : 0)){
        case GEOAC_EQ_GLOBAL * 2 + 1: hipLaunchKernelGGL(k_ppfix<EqGlobal<true>>, gf, b, 0, s, *P); break;
        case GEOAC_EQ_GLOBAL * 2 + 0: hipLaunchKernelGGL(k_ppfix<EqGlobal<false>>, gf, b, 0, s, *P); break;
        case GEOAC_EQ_3D * 2 + 1:     hipLaunchKernelGGL(k_ppfix<Eq3D<true>>, gf, b, 0, s, *P); break;
        case GEOAC_EQ_3D * 2 + 0:     hipLaunchKernelGGL(k_ppfix<Eq3D<false>>, gf, b, 0, s, *P); break;
        case GEOAC_EQ_2D * 2 + 1:     hipLaunchKernelGGL(k_ppfix<Eq2D<true>>, gf, b, 0, s, *P); break;
        case GEOAC_EQ_2D * 2 + 0:     hipLaunchKernelGGL(k_ppfix<Eq2D<false>>, gf, b, 0, s, *P); break;
        default: break;
    }
    return hipGetLastError();
}

extern "C" hipError_t geoac_launch_atab_build(const GeoacDevParams* P, double* tab, double tol, hipStream_t s){
    hipLaunchKernelGGL(k_atab_build, dim3((P->nseg + 2 + GEOAC_LAT_N + 63) / 64), dim3(64), 0, s, *P, tab, tol);
    return hipGetLastError();
}

extern "C" hipError_t geoac_launch_probe_atmo1d(const GeoacDevParams* P, int n, const double* x, double* out9, double* rho, hipStream_t s){
    hipLaunchKernelGGL(k_probe_atmo1d, dim3((n + 255) / 256), dim3(256), 0, s, *P, n, x, out9, rho);
    return hipGetLastError();
}
extern "C" hipError_t geoac_launch_probe_absorption(const GeoacDevParams* P, int n, const double* x, const double* f, double* out, hipStream_t s){
    hipLaunchKernelGGL(k_probe_absorption, dim3((n + 255) / 256), dim3(256), 0, s, *P, n, x, f, out);
    return hipGetLastError();
}
extern "C" hipError_t geoac_launch_probe_atab(const GeoacDevParams* P, int n, const double* x, double* out, hipStream_t s){
    if(!P->atab) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_probe_atab, dim3((n + 255) / 256), dim3(256), 0, s, *P, n, x, out);
    return hipGetLastError();
}
extern "C" int geoac_kernels_cart_rec(void){ return GRec<false>::N; }     // doubles per record of the table the Cartesian grid kernels read
extern "C" hipError_t geoac_launch_probe_grid(const GeoacDevParams* P, int n, const double* a0, const double* a1, const double* a2, int coop,
                                              double* out30, double* api7, hipStream_t s){
    const size_t lds = coop ? (size_t)(GEOAC_GLDS_BYTES > 64 * GEOAC_COOP_SLOT ? GEOAC_GLDS_BYTES : 64 * GEOAC_COOP_SLOT) : 0;
    if(P->eqset == GEOAC_EQ_GLOBAL_RNGDEP) hipLaunchKernelGGL(k_probe_grid<true>, dim3((n + 63) / 64), dim3(64), lds, s, *P, n, a0, a1, a2, coop, out30, api7);
    else                                   hipLaunchKernelGGL(k_probe_grid<false>, dim3((n + 63) / 64), dim3(64), lds, s, *P, n, a0, a1, a2, coop, out30, api7);
    return hipGetLastError();
}

extern "C" hipError_t geoac_launch_compact(const GeoacDevParams* P, const int* cur, const int* n_cur, int n_first, int* next, int* n_next, hipStream_t s){
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, s, *P, cur, n_cur, n_first, next, n_next);
    return hipGetLastError();
}

extern "C" hipError_t geoac_launch_arrival(const GeoacDevParams* P, hipStream_t s){
    dim3 b(256), g((unsigned)(((long long)P->n_rays * (P->bounces + 1) + 255) / 256));
    const bool amp = P->calc_amp != 0;
    switch(P->eqset){
        case GEOAC_EQ_GLOBAL: if(amp) hipLaunchKernelGGL(k_arrival<EqGlobal<true>>, g, b, 0, s, *P); else hipLaunchKernelGGL(k_arrival<EqGlobal<false>>, g, b, 0, s, *P); break;
        case GEOAC_EQ_3D:     if(amp) hipLaunchKernelGGL(k_arrival<Eq3D<true>>, g, b, 0, s, *P);     else hipLaunchKernelGGL(k_arrival<Eq3D<false>>, g, b, 0, s, *P); break;
        case GEOAC_EQ_2D:     if(amp) hipLaunchKernelGGL(k_arrival<Eq2D<true>>, g, b, 0, s, *P);     else hipLaunchKernelGGL(k_arrival<Eq2D<false>>, g, b, 0, s, *P); break;
        case GEOAC_EQ_3D_RNGDEP:     if(amp) hipLaunchKernelGGL(k_arrival<Eq3DRngDep<true>>, g, b, 0, s, *P);     else hipLaunchKernelGGL(k_arrival<Eq3DRngDep<false>>, g, b, 0, s, *P); break;
        case GEOAC_EQ_GLOBAL_RNGDEP: if(amp) hipLaunchKernelGGL(k_arrival<EqGlobalRngDep<true>>, g, b, 0, s, *P); else hipLaunchKernelGGL(k_arrival<EqGlobalRngDep<false>>, g, b, 0, s, *P); break;
        default: return hipSuccess;
    }
    return hipGetLastError();
}
extern "C" hipError_t geoac_launch_accum(const GeoacDevParams* P, hipStream_t s){
    dim3 b(256), g2((P->n_cols_bound + 255) / 256);
    hipLaunchKernelGGL(k_accum, g2, b, 0, s, *P);
    return hipGetLastError();
}
#endif  // GEOAC_NO_LAUNCHERS
