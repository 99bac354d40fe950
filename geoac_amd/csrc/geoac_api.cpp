// geoac_api.cpp - the C ABI of include/geoac_hip.h: context management, table building, the epoch loop.
// Compiled with hipcc together with geoac_kernels.hip into libgeoac_hip.so.  No CPU compute fallback
// exists here: every compute entry point needs a HIP device and fails loudly without one.
#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <errno.h>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/geoac_hip.h"
#include "../../include/geoac_probe.h"
#include "../../include/geoac_host.h"
#include "geoac_device.h"

extern "C" void geoac_natural_spline_slopes(int n, const double* x, const double* f, double* slopes);
extern "C" hipError_t geoac_launch_init(const GeoacDevParams* P, hipStream_t s);
extern "C" hipError_t geoac_launch_rk4(const GeoacDevParams* P, int block, hipStream_t s, unsigned* n_wg);
extern "C" size_t geoac_duo_lds(int nseg);
extern "C" size_t geoac_trio_lds(int nseg);
extern "C" int geoac_build_has_ab(void);
extern "C" size_t geoac_gridbuild_work_doubles(int nx, int ny, int nz);
extern "C" size_t geoac_gridpack_doubles(int nx, int ny, int nz);
extern "C" hipError_t geoac_gridpack_launch(int nx, int ny, int nz, const double* d_tab, double* d_tab8, hipStream_t s);
extern "C" int geoac_kernels_cart_rec(void);
extern "C" hipError_t geoac_gridbuild_launch(int glob, int nx, int ny, int nz, const double* d_x, const double* d_y, const double* d_z,
                                             const double* d_fields, double* d_work, double* d_tab, hipStream_t s);
extern "C" hipError_t geoac_launch_accum(const GeoacDevParams* P, hipStream_t s);
extern "C" hipError_t geoac_launch_arrival(const GeoacDevParams* P, hipStream_t s);
extern "C" hipError_t geoac_launch_compact(const GeoacDevParams* P, const int* cur, const int* n_cur, int n_first, int* next, int* n_next, hipStream_t s);
extern "C" hipError_t geoac_launch_probe_atmo1d(const GeoacDevParams* P, int n, const double* x, double* out9, double* rho, hipStream_t s);
extern "C" hipError_t geoac_launch_probe_absorption(const GeoacDevParams* P, int n, const double* x, const double* f, double* out, hipStream_t s);
extern "C" hipError_t geoac_launch_probe_atab(const GeoacDevParams* P, int n, const double* x, double* out, hipStream_t s);
extern "C" hipError_t geoac_launch_probe_grid(const GeoacDevParams* P, int n, const double* a0, const double* a1, const double* a2, int coop,
                                              double* out30, double* api7, hipStream_t s);
extern "C" hipError_t geoac_launch_gate(const GeoacDevParams* P, unsigned long long expected, hipStream_t s);
extern "C" hipError_t geoac_launch_postpass(const GeoacDevParams* P, int rows, hipStream_t s);
extern "C" hipError_t geoac_launch_postpass_tab(const GeoacDevParams* P, int rows, hipStream_t s);
extern "C" hipError_t geoac_launch_atab_build(const GeoacDevParams* P, double* tab, double tol, hipStream_t s);

namespace {

const double kPi = 3.141592653589793238462643;
const double kGam = 1.4, kRgas = 287.05, kGamR = 0.00040187;

// A buffer that has to grow is not freed on the spot when it is small: hipFree waits for the whole device, i.e. for the launches of the OTHER
// contexts of an eigenray search round (a 4-ray refinement group measured at 324 instead of 190 ms when its context's buffers had to grow
// beside the other groups' launches).  The old allocation goes to a process-wide list that is emptied when a context is destroyed or when the
// list holds more than 256 MB.  Large buffers (the path chunks of a big fan) are freed at once, as before: their memory is needed.
struct DeferredFrees {
    std::mutex mu; std::vector<void*> ptrs; size_t bytes = 0;
    void add(void* p, size_t n){
        std::vector<void*> now;
        {   std::lock_guard<std::mutex> lk(mu);
            ptrs.push_back(p); bytes += n;
            if(bytes > (256ull << 20)){ now.swap(ptrs); bytes = 0; } }
        for(void* q : now) hipFree(q);
    }
    void drain(){
        std::vector<void*> now;
        {   std::lock_guard<std::mutex> lk(mu); now.swap(ptrs); bytes = 0; }
        for(void* q : now) hipFree(q);
    }
    // the last context to leave geoac_fan_launch: no launch of this library is in flight, hipFree has nothing of ours to wait for (a long-lived single context -
    // bench.py, a pool - would otherwise keep up to 256 MB of outgrown buffers until it is destroyed)
    void drain_if_any(){
        bool any;
        {   std::lock_guard<std::mutex> lk(mu); any = !ptrs.empty(); }
        if(any) drain();
    }
};
DeferredFrees g_deferred;
std::atomic<int> g_launching{0};       // contexts inside geoac_fan_launch: the last one out empties the deferred list (nobody's launches left for hipFree to wait on)

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    bool owned = true;                 // false: a view of another context's buffer (geoac_clone: the atmosphere tables are shared, read-only)
    hipError_t ensure(size_t need){
        if(need <= bytes && p) return hipSuccess;
        if(p && owned){ if(bytes <= (32ull << 20)) g_deferred.add(p, bytes); else hipFree(p); }
        p = nullptr; bytes = 0; owned = true;
        hipError_t e = hipMalloc(&p, need);
        if(e == hipSuccess) bytes = need;
        return e;
    }
    void release(){ if(p && owned) hipFree(p); p = nullptr; bytes = 0; owned = true; }
    void view_of(const DevBuf& o){ release(); p = o.p; bytes = o.bytes; owned = false; }
};

}  // namespace

struct geoac_ctx {
    int eqset = 0, device = 0;
    GeoacDevParams lastP{};          // parameter block of the last completed launch (device-function probes, include/geoac_probe.h)
    int  accum_batch = -1;           // ACCUM_BATCH: k_accum fetches eight rows' contributions together; < 0: in the late epochs (few waves alive)
    int  chunk_gib = 40;             // CHUNK_GIB: largest path chunk (three of them + their contribution buffers at 1/3 of that each: <= 160 GiB of the 288 by default)
    int  cu_split = -1;              // CU_SPLIT: compute units of the device set aside for the post-pass (streams with a CU mask, hipExtStreamCreateWithCUMask): the RK4 launches run on the others;
                                     // 0: no partition (the post-pass goes where the RK4 workgroups leave room); < 0: by launch plan
    int  cu_split_made = 0;          // ... the split the two masked streams below were created for (0: none yet)
    hipStream_t rk4_cu_stream = nullptr, pp_cu_stream = nullptr;
    hipEvent_t ev_cu = nullptr;      // orders the masked RK4 stream behind the context's own
    int  pp_lds_pad = -1;            // PP_LDS_PAD: bytes of LDS a table post-pass workgroup asks for (its occupancy knob: 160 KiB per CU / this = workgroups of four waves per CU); < 0: by launch plan
    int  pp_lds_table = -1;          // table post-pass of the spherical set: the table entry in LDS, 127 registers (PP_LDS_TABLE=1; default off: no faster, geoac_fan_launch)
    // atmosphere generation, shared with the clones of this context (geoac_clone: they hold VIEWS of its tables): bumped by every upload, set to ~0 when the context is
    // destroyed.  A clone remembers the value it was made at and refuses to launch once it differs - its views would dangle or show another atmosphere.
    std::shared_ptr<std::atomic<unsigned long long>> atmo_gen = std::make_shared<std::atomic<unsigned long long>>(1);
    std::shared_ptr<std::atomic<unsigned long long>> src_gen;      // clones: the source's generation counter ...
    unsigned long long src_gen_at_clone = 0;                       // ... and its value when the clone was made
    unsigned long long sticky_flags = 0;   // GEOAC_FAN_*_FALLBACK: plan features this context has withdrawn after a failed attempt (geoac_fan_status)
    int  launch_repeats = 0;         // fans that were run a second time for that reason
    bool sub_test_stall = false;     // SUB_TEST_STALL (tests): the cooperative grid kernels' workgroups of sub-epoch 0 do not publish their flag - forces the hand-off time-out
    int  pp_onetrip = -1;            // table post-pass: record + table entry of the neighbouring segment in one trip (PP_ONETRIP; < 0: every fan but the hybrid ones)
    bool tile_rays = true;           // grid sets: Z-order over (inclination, azimuth) ranks instead of the inclination order (GEOAC_TILE=0)
    bool sort_rays = true;           // integrate the rays in order of launch inclination, results in caller order (GEOAC_SORT=0: caller order).
                                     // Ray length is mostly a function of inclination (ground-hugging rays take 1 m steps), so whole waves finish early
                                     // instead of every wave waiting for its one long ray: the post-pass then lands on idle SIMDs (metric fan +13 %)
    DevBuf perm; bool have_perm = false;
    bool no_quad = false;            // GEOAC_NO_QUAD=1: never use the multi-lane grid kernels
    bool hex = true;                 // HEX=0: the smallest spherical grid fans with amplitudes on the eight-lane kernel instead of the sixteen-lane one
    bool oct = true;                 // GEOAC_OCT=0: small spherical grid fans with amplitudes on the four-lane kernel instead of the eight-lane one
    int  grid_lanes = 0;             // GEOAC_GRID_LANES=1|2|4: force the lanes-per-ray variant of the grid kernels (tests); 0 = by fan size
    int  spread_override = 0;        // GEOAC_SPREAD=n: force n-way lane thinning of the grid-set RK4 waves (1 = dense); 0 = automatic
    bool compact = true;             // GEOAC_COMPACT=0: every epoch runs over all slots (no live-ray compaction between epochs)
    DevBuf colmap[3], ncols;         // per chunk: column -> slot list of the rays alive at the start of that epoch; their counts (3 ints)
    bool quad_cache = true;          // GEOAC_QUAD_CACHE=0: four-lane grid kernels without the per-lane record cache (A/B)
    int  sub_min_waves = 1024;       // GEOAC_SUB_MIN_WAVES: smallest launch (waves) that is cut into sub-epochs (tests lower it to cover the hand-off with small fans)
    int  sub_epochs = 4;             // GEOAC_SUB_EPOCHS: sub-epochs of the cooperative grid kernels when a fan has more waves than the chip has wave slots (k_rk4); 1 = off
    DevBuf sub_flags;
    bool grid_coop = true;           // GEOAC_GRID_COOP=0: per-lane table gathers instead of the wave-cooperative gather (A/B runs, schedule-independence test)
    hipStream_t stream = nullptr; bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> evs;                  // per-epoch markers: [4e], [4e+1] around k_rk4 (ctx stream); [4e+2], [4e+3] around the post-pass (pp stream)
    geoac_params prm{};
    bool have_params = false, have_atmo = false, have_angles = false, ran = false;
    // host copy of the 1-D atmosphere (for the SuthBass reference state)
    std::vector<double> x, T, u, v, rho, sl;
    int n_nodes = 0;
    // range-dependent grid
    int gnx = 0, gny = 0;
    std::vector<double> gx, gy;
    DevBuf d_gx, d_gy, d_gz, d_gtab, d_gtab8, d_consts;      // d_gtab8: packed records the Cartesian kernels read (geoac_rngdep.h); d_gtab: the full table
    size_t gtab_bytes = 0;           // of the table the kernels read
    bool have_grid = false;
    // device
    DevBuf seg, rhot, theta, phi, state, rec, counters;
    DevBuf path[3], contrib[3], nrows[3], legend[3], nlegend[3];   // epoch chunks, two or three in rotation (RK4 of epoch e+1 overlaps the post-pass of e)
    hipStream_t pp_stream = nullptr;                                // post-pass stream
    hipStream_t acc_stream = nullptr;                               // k_accum stream
    std::vector<hipEvent_t> evp;                                    // per epoch: k_postpass has finished (k_accum may start)
    hipStream_t rk4b_stream = nullptr;                              // second RK4 stream (hybrid fans: the one-lane launch)
    std::vector<hipEvent_t> evj;                                    // per epoch: [2e] the second RK4 launch has finished, [2e+1] live-ray count on the host
    DevBuf ev_row[3], ev_m[3], ev_amp[3], nev[3], smp_out;          // WriteRays / WriteCaustics events and the sample list
    long long smp_cap = 4ll << 20;                                  // sample records the device list can hold (GEOAC_SMP_CAP)
    unsigned long long n_samples = 0;
    unsigned long long* h_counters = nullptr;     // pinned
    int n_rays = 0, n_pad = 0, legs = 0;
    // last launch
    double ms_total = 0, ms_rk4 = 0, ms_post = 0;
    unsigned long long n_epochs = 0, path_bytes_w = 0, path_bytes_r = 0, total_steps = 0, err_flags = 0;
    int s_rows_override = 0;
    bool no_overlap = false;                      // GEOAC_NO_OVERLAP=1: post-pass on the RK4 stream (diagnostics)
    int pp_blocks = 0;                            // GEOAC_PP_BLOCKS: post-pass grid (256-thread blocks); 0 = one block per 256 segments.
                                                  // Measured: a short full-occupancy burst disturbs k_rk4 LESS than a long thin sweep
                                                  // (avg k_rk4 launch 10.9 ms vs 11.7-12.8 ms at 8192-256 blocks).
    double pair_frac = -1.0;                      // hybrid fans: share of the inclination-sorted rays that get two lanes (PAIR_FRAC; >= 1: all; < 0: the set's own default, plan_pair_frac)
    double stagger_frac = -1.0;                   // staggered epochs (fans of the stratified sets with more waves than the chip has wave slots): share of the inclination-sorted rays - the shallow,
                                                  // long ones - that get a whole epoch's rows per launch while the others get stagger_rows of them (STAGGER_FRAC; 0: off; < 0: by launch plan)
    double stagger_rows = -1.0;                   // ... STAGGER_ROWS (< 0: by launch plan)
    double hybrid_rows = -1.0;                    // hybrid fans: rows per epoch of the one-lane launch relative to the two-lane launch (HYBRID_ROWS; < 0: the set's own default, plan_hybrid_rows)
    bool two_chunks = false;                      // GEOAC_TWO_CHUNKS=1: two path chunks in rotation instead of three (A/B measurements)
    bool trace_epochs = false;                    // GEOAC_TRACE_EPOCHS=1: per-epoch live counts on stderr
    bool no_gate = false;                         // GEOAC_NO_GATE=1: post-pass not held back behind the next RK4 launch (A/B measurements)
    bool no_pair = false;                         // GEOAC_NO_PAIR=1: force one lane per ray (A/B measurements)
    bool grid_build_host = false;                 // GRID_BUILD=host: evaluation table of the grid sets by the host twin (geoac_grid_table_eq) instead of on the device
    int ev_slack = 72;                            // GEOAC_EV_SLACK: per-epoch event rows of a ray beyond its raypath samples (caustics); tests lower it to reach the overflow path
    int trio = 0;                                 // TRIO=1: slots that would take two lanes per ray run the wave-specialised kernel k_rk4_trio (the ray on one wave, one launch-angle system on each of two more)
    int duo = 0;                                  // GEOAC_DUO=1: the wave-specialised kernel k_rk4_duo for Global fans with amplitudes (measured SLOWER than the two-lane
                                                  // kernel on MI355X - 3.0 vs 2.7 us per step, DESIGN 3 - kept for A/B runs and the schedule-independence tests); 32, 66: timing diagnostics
    // absorption table of the stratified sets (k_atab_build): rebuilt when the atmosphere or one of the parameters it depends on changes
    bool abs_table = true;                        // GEOAC_ABS_TABLE=0: exact Sutherland-Bass evaluation at every segment midpoint (A/B runs, equivalence test)
    DevBuf atab;
    DevBuf ppfix;                                 // fix-up list of the table post-pass (k_postpass_tab -> k_ppfix)
    double atab_tol = 1e-10;                      // ABS_TABLE_TOL: relative error at its check points above which a table entry is flagged
    int ppfix_cap = 1 << 20;                      // PPFIX_CAP: segments per epoch the table may leave to k_ppfix (8 MB); more: the fan is repeated with the exact post-pass
    unsigned long long atmo_version = 0, atab_version = ~0ull;
    double atab_key[7] = {0, 0, 0, 0, 0, 0, 0};   // freq, tweak_abs, T_o, P_o, r_earth, strip width, tolerance
    int atab_entries = 0, atab_flagged = 0;       // of the current table
    double atab_worst = 0;                        // largest relative error the build saw at its check points, unflagged entries
    unsigned long long pp_fixup_segments = 0;     // last launch: path segments the table did not serve (evaluated exactly)
    std::string err;
};

namespace {

int fail(geoac_ctx* c, int code, const std::string& msg){ if(c) c->err = msg; return code; }
int hipfail(geoac_ctx* c, hipError_t e, const char* where){
    return fail(c, GEOAC_E_HIP, std::string(where) + ": " + hipGetErrorString(e));
}
#define HIPCHK(call) do { hipError_t e_ = (call); if(e_ != hipSuccess) return hipfail(ctx, e_, #call); } while(0)

// value of the reference-form cubic at node arrays (host; used only for the SuthBass ground state)
double host_spline_f(const std::vector<double>& xv, const std::vector<double>& fv, const double* sl, double x){
    int n = (int)xv.size();
    if(x > xv[n - 1]) x = xv[n - 1];
    if(x < xv[0]) x = xv[0];
    int k = 0;
    while(k < n - 2 && x > xv[k + 1]) k++;
    double h = xv[k + 1] - xv[k];
    double X = (x - xv[k]) / h;
    double df = fv[k + 1] - fv[k];
    double A = sl[k] * h - df, B = -sl[k + 1] * h + df;
    return (1.0 - X) * fv[k] + X * fv[k + 1] + X * (1.0 - X) * (A * (1.0 - X) + B * X);
}

// cubic of one spline segment in powers of t = x - x_k: geoac_spline_segment_cubic (geoac_host.cpp)
static inline void seg_coeffs(double x0, double x1, double f0, double f1, double s0, double s1, double* c, bool deriv_form){
    geoac_spline_segment_cubic(x0, x1, f0, f1, s0, s1, c, deriv_form ? 1 : 0);
}

}  // namespace

extern "C" {

const char* geoac_version(void){ return "geoac_hip 0.1 (gfx950)"; }

const char* geoac_strerror(int code){
    switch(code){
        case GEOAC_OK: return "ok";
        case GEOAC_E_INVALID: return "invalid argument or call order";
        case GEOAC_E_NODEVICE: return "no usable HIP device (this library has no CPU fallback)";
        case GEOAC_E_HIP: return "HIP runtime error";
        case GEOAC_E_UNSUPPORTED: return "equation set or mode not implemented";
        case GEOAC_E_CAPACITY: return "capacity exceeded (step_limit or buffer)";
        case GEOAC_E_NOMEM: return "out of memory";
        default: return "unknown error";
    }
}

const char* geoac_last_error(geoac_ctx* ctx){ return ctx ? ctx->err.c_str() : ""; }

int geoac_default_params(int eqset, geoac_params* p){
    if(!p) return GEOAC_E_INVALID;
    memset(p, 0, sizeof(*p));
    p->ds_min = 0.001; p->ds_max = 0.5;                         // GeoAc.Parameters.cpp:19-20
    if(eqset == GEOAC_EQ_GLOBAL || eqset == GEOAC_EQ_GLOBAL_RNGDEP){
        p->ray_limit = 10000.0;                                 // GeoAc.Parameters.Global.cpp:23
        p->range_limit = 1500.0;                                // G2S_GlobalSpline1D.cpp:27
        p->r_earth = 6370.0;                                    // G2S_GlobalSpline1D.cpp:35
        p->src[0] = 0.0; p->src[1] = 30.0; p->src[2] = 0.0;     // GeoAcGlobal_main.cpp:120
    } else {
        p->ray_limit = 5000.0;                                  // GeoAc.Parameters.cpp:23
        p->range_limit = 10000.0;                               // G2S_Spline1D.cpp:27
        p->r_earth = 0.0;
    }
    p->vert_limit = NAN;                                        // = top of the profile, set at upload (GeoAc_SetPropRegion)
    p->z_grnd = 0.0; p->tweak_abs = 0.3; p->freq = 0.1;
    p->bounces = 2; p->calc_amp = 1; p->mode = 0; p->sample_stride = 25;
    for(int q = 0; q < 4; q++) p->xy_limits[q] = NAN;           // RngDep: grid extents at upload (GeoAc_SetPropRegion)
    return GEOAC_OK;
}

// ---- launch-plan options (A/B measurements, tests; results never depend on them) ----
static const char* const kOptionNames[] = {
    "S_ROWS", "NO_OVERLAP", "PP_BLOCKS", "ABS_TABLE", "ABS_TABLE_TOL", "PPFIX_CAP", "DUO", "TRIO", "EV_SLACK", "NO_PAIR", "PAIR_FRAC", "HYBRID_ROWS", "STAGGER_FRAC", "STAGGER_ROWS", "TWO_CHUNKS", "TRACE_EPOCHS", "NO_GATE", "SORT", "TILE", "PP_ONETRIP", "PP_LDS_TABLE", "PP_LDS_PAD", "CU_SPLIT", "CHUNK_GIB", "ACCUM_BATCH",
    "NO_QUAD", "GRID_LANES", "OCT", "HEX", "SPREAD", "COMPACT", "QUAD_CACHE", "GRID_COOP", "SUB_EPOCHS", "SUB_MIN_WAVES", "SUB_TEST_STALL", "SMP_CAP", "GRID_BUILD", nullptr };
const char* const* geoac_option_names(void){ return kOptionNames; }

int geoac_set_option(geoac_ctx* ctx, const char* key, const char* value){
    if(!ctx || !key || !value) return GEOAC_E_INVALID;
    std::string k(key);
    if(k.rfind("GEOAC_", 0) == 0) k = k.substr(6);
    for(char& c : k) c = (char)toupper((unsigned char)c);
    // A value that does not parse, or lies outside what the knob accepts, is an error (GEOAC_E_INVALID + message) - never a silent 0 or a silently
    // ignored setting: an A/B run or a schedule-independence test must not measure the default plan while believing a knob was applied
    bool int_ok = false, dbl_ok = false;
    long long lv = 0; double dv = 0.0;
    { char* end = nullptr; errno = 0; lv = strtoll(value, &end, 10); int_ok = (end != value && *end == 0 && errno == 0); }
    { char* end = nullptr; errno = 0; dv = strtod(value, &end); dbl_ok = (end != value && *end == 0 && errno == 0 && dv == dv); }
    const int iv = (int)lv;
    auto bad = [&](const char* want){ return fail(ctx, GEOAC_E_INVALID, "set_option: " + k + "=" + value + ": expected " + want); };
    auto flag = [&](bool& dst) -> int { if(!int_ok || (lv != 0 && lv != 1)) return bad("0 or 1"); dst = lv != 0; return GEOAC_OK; };
    if(k == "S_ROWS"){ if(!int_ok || lv < 0 || lv > 0x7fffffff) return bad("a row count >= 0 (0: sized from the free memory)"); ctx->s_rows_override = iv; }
    else if(k == "NO_OVERLAP") return flag(ctx->no_overlap);
    else if(k == "PP_BLOCKS"){ if(!int_ok || lv < 0 || lv > 0x7fffffff) return bad("a block count >= 0"); ctx->pp_blocks = iv; }
    else if(k == "ABS_TABLE") return flag(ctx->abs_table);
    else if(k == "ABS_TABLE_TOL"){ if(!dbl_ok || !(dv > 0.0)) return bad("a tolerance > 0"); ctx->atab_tol = dv; }
    else if(k == "PPFIX_CAP"){ if(!int_ok || lv <= 0 || lv > 0x3fffffff) return bad("a capacity > 0"); ctx->ppfix_cap = iv; }
    else if(k == "DUO"){
        if(!int_ok || lv < 0 || lv > 3) return bad("0..3");
        if(lv && !geoac_build_has_ab()) return fail(ctx, GEOAC_E_UNSUPPORTED, "set_option: DUO: the wave-specialised kernel is part of A/B builds only (make AB=1)");
        ctx->duo = iv;
    }
    else if(k == "TRIO"){
        if(!int_ok || lv < 0 || lv > 31) return bad("0, 1 (3, 5, 11, 27: timing diagnostics)");
        if(lv && !geoac_build_has_ab()) return fail(ctx, GEOAC_E_UNSUPPORTED, "set_option: TRIO: the three-wave kernel is part of A/B builds only (make AB=1)");
        ctx->trio = iv;
    }
    else if(k == "EV_SLACK"){ if(!int_ok || lv < 0 || lv > 1000000) return bad("a slot count >= 0"); ctx->ev_slack = iv; }
    else if(k == "NO_PAIR") return flag(ctx->no_pair);
    else if(k == "PAIR_FRAC"){ if(!dbl_ok || dv < 0.0 || dv > 1.0) return bad("a share in [0, 1]"); ctx->pair_frac = dv; }
    else if(k == "STAGGER_FRAC"){ if(!dbl_ok || dv < -1.0 || dv > 0.9) return bad("a share in [0, 0.9] (0: off), or -1 (by launch plan)"); ctx->stagger_frac = dv; }
    else if(k == "STAGGER_ROWS"){ if(!dbl_ok || !(dv > 0.0 && dv <= 1.0)) return bad("a ratio in (0, 1]"); ctx->stagger_rows = dv; }
    else if(k == "HYBRID_ROWS"){ if(!dbl_ok || !(dv > 0.0 && dv <= 1.0)) return bad("a ratio in (0, 1]"); ctx->hybrid_rows = dv; }
    else if(k == "TWO_CHUNKS") return flag(ctx->two_chunks);
    else if(k == "TRACE_EPOCHS") return flag(ctx->trace_epochs);
    else if(k == "NO_GATE") return flag(ctx->no_gate);
    else if(k == "SORT") return flag(ctx->sort_rays);
    else if(k == "TILE") return flag(ctx->tile_rays);
    else if(k == "PP_ONETRIP"){ if(!int_ok || lv < -1 || lv > 1) return bad("0, 1 or -1 (by launch plan)"); ctx->pp_onetrip = iv; }
    else if(k == "CHUNK_GIB"){ if(!int_ok || lv < 1 || lv > 256) return bad("GiB per path chunk in 1 .. 256"); ctx->chunk_gib = iv; }
    else if(k == "ACCUM_BATCH"){ if(!int_ok || lv < -1 || lv > 1) return bad("0, 1 or -1 (by launch plan: the late epochs)"); ctx->accum_batch = iv; }
    else if(k == "CU_SPLIT"){ if(!int_ok || lv < -1 || lv > 248) return bad("compute units for the post-pass, 0 .. 248 (0: no partition), or -1 (by launch plan)"); ctx->cu_split = iv; }
    else if(k == "PP_LDS_PAD"){ if(!int_ok || lv < -1 || lv > 160 * 1024) return bad("bytes of LDS in 0 .. 163840, or -1 (by launch plan)"); ctx->pp_lds_pad = iv; }
    else if(k == "PP_LDS_TABLE"){ if(!int_ok || lv < -1 || lv > 1) return bad("0, 1 or -1 (by launch plan)"); ctx->pp_lds_table = iv; }
    else if(k == "NO_QUAD") return flag(ctx->no_quad);
    else if(k == "GRID_LANES"){
        if(!int_ok || !(iv == 0 || iv == 1 || iv == 2 || iv == 4 || iv == 8 || iv == 16)) return bad("0, 1, 2, 4, 8 or 16");
        if(iv == 2 && !geoac_build_has_ab()) return fail(ctx, GEOAC_E_UNSUPPORTED, "set_option: GRID_LANES=2: the grid sets' two-lane kernels are part of A/B builds only (make AB=1)");
        ctx->grid_lanes = iv;
    }
    else if(k == "OCT") return flag(ctx->oct);
    else if(k == "HEX") return flag(ctx->hex);
    else if(k == "SPREAD"){ if(!int_ok || lv < 0 || lv > 64 || (lv & (lv - 1))) return bad("0 (by launch plan) or a power of two up to 64"); ctx->spread_override = iv; }
    else if(k == "COMPACT") return flag(ctx->compact);
    else if(k == "QUAD_CACHE") return flag(ctx->quad_cache);
    else if(k == "GRID_COOP") return flag(ctx->grid_coop);
    else if(k == "SUB_EPOCHS"){ if(!int_ok || lv < 1 || lv > 16) return bad("1..16"); ctx->sub_epochs = iv; }
    else if(k == "SUB_MIN_WAVES"){ if(!int_ok || lv < 0 || lv > 0x7fffffff) return bad("a wave count >= 0"); ctx->sub_min_waves = iv; }
    else if(k == "SUB_TEST_STALL") return flag(ctx->sub_test_stall);
    else if(k == "SMP_CAP"){ if(!int_ok || lv <= 0) return bad("a row capacity > 0"); ctx->smp_cap = lv; }
    else if(k == "GRID_BUILD"){ if(strcmp(value, "host") != 0 && strcmp(value, "device") != 0) return bad("host or device"); ctx->grid_build_host = (strcmp(value, "host") == 0); }
    else return fail(ctx, GEOAC_E_INVALID, "set_option: unknown key " + k);
    return GEOAC_OK;
}

int geoac_create(geoac_ctx** out, int eqset, int device){
    if(!out) return GEOAC_E_INVALID;
    *out = nullptr;
    if(eqset < GEOAC_EQ_2D || eqset > GEOAC_EQ_GLOBAL_RNGDEP) return GEOAC_E_INVALID;
    int ndev = 0;
    if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GEOAC_E_NODEVICE;
    if(device < 0 || device >= ndev) return GEOAC_E_NODEVICE;
    if(hipSetDevice(device) != hipSuccess) return GEOAC_E_NODEVICE;
    geoac_ctx* ctx = new geoac_ctx();
    ctx->eqset = eqset; ctx->device = device;
    // RK4 launches must win the CUs at every epoch boundary: the post-pass of the previous epoch becomes runnable at the same
    // moment and its short workgroups would otherwise occupy the CUs the RK4 workgroups (153 KB of LDS each) need
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);          // lo = least urgent (numerically largest)
    // (a failure part-way leaves streams / events behind: geoac_destroy releases whatever exists)
    if(hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_hi) != hipSuccess){ ctx->stream = nullptr; geoac_destroy(ctx); return GEOAC_E_HIP; }
    ctx->own_stream = true;
    if(hipStreamCreateWithPriority(&ctx->pp_stream, hipStreamNonBlocking, prio_lo) != hipSuccess){ ctx->pp_stream = nullptr; geoac_destroy(ctx); return GEOAC_E_HIP; }
    if(hipStreamCreateWithPriority(&ctx->acc_stream, hipStreamNonBlocking, prio_lo) != hipSuccess){ ctx->acc_stream = nullptr; geoac_destroy(ctx); return GEOAC_E_HIP; }
    if(hipStreamCreateWithPriority(&ctx->rk4b_stream, hipStreamNonBlocking, prio_hi) != hipSuccess){ ctx->rk4b_stream = nullptr; geoac_destroy(ctx); return GEOAC_E_HIP; }
    if(hipEventCreate(&ctx->ev0) != hipSuccess){ ctx->ev0 = nullptr; geoac_destroy(ctx); return GEOAC_E_HIP; }
    if(hipEventCreate(&ctx->ev1) != hipSuccess){ ctx->ev1 = nullptr; geoac_destroy(ctx); return GEOAC_E_HIP; }
    if(hipHostMalloc((void**)&ctx->h_counters, 32 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess){
        ctx->h_counters = nullptr; geoac_destroy(ctx); return GEOAC_E_HIP;
    }
    geoac_default_params(eqset, &ctx->prm);
    // launch-plan options come through geoac_set_option; the environment is read only when GEOAC_DEBUG_ENV=1 (the drivers' A/B runs)
    const char* dbg = getenv("GEOAC_DEBUG_ENV");
    if(dbg && atoi(dbg) != 0){
        for(const char* const* k = geoac_option_names(); *k; k++){
            const std::string name = std::string("GEOAC_") + *k;
            const char* v = getenv(name.c_str());
            if(v) geoac_set_option(ctx, *k, v);
        }
    }
    *out = ctx;
    return GEOAC_OK;
}

int geoac_destroy(geoac_ctx* ctx){
    if(!ctx) return GEOAC_E_INVALID;
    ctx->atmo_gen->store(~0ull);                                  // (clones of this context must not launch on its freed tables)
    hipSetDevice(ctx->device);
    if(ctx->stream) hipStreamSynchronize(ctx->stream);
    DevBuf* bufs[] = { &ctx->seg, &ctx->rhot, &ctx->theta, &ctx->phi, &ctx->state, &ctx->rec, &ctx->counters, &ctx->perm,
                       &ctx->path[0], &ctx->path[1], &ctx->path[2], &ctx->contrib[0], &ctx->contrib[1], &ctx->contrib[2],
                       &ctx->nrows[0], &ctx->nrows[1], &ctx->nrows[2], &ctx->legend[0], &ctx->legend[1], &ctx->legend[2],
                       &ctx->nlegend[0], &ctx->nlegend[1], &ctx->nlegend[2],
                       &ctx->ev_row[0], &ctx->ev_row[1], &ctx->ev_row[2], &ctx->ev_m[0], &ctx->ev_m[1], &ctx->ev_m[2],
                       &ctx->ev_amp[0], &ctx->ev_amp[1], &ctx->ev_amp[2], &ctx->nev[0], &ctx->nev[1], &ctx->nev[2], &ctx->smp_out,
                       &ctx->d_gx, &ctx->d_gy, &ctx->d_gz, &ctx->d_gtab, &ctx->d_gtab8, &ctx->d_consts, &ctx->sub_flags, &ctx->colmap[0], &ctx->colmap[1], &ctx->colmap[2], &ctx->ncols, &ctx->atab, &ctx->ppfix };
    for(DevBuf* b : bufs) b->release();
    g_deferred.drain();
    if(ctx->h_counters) hipHostFree(ctx->h_counters);
    if(ctx->ev0) hipEventDestroy(ctx->ev0);
    if(ctx->ev1) hipEventDestroy(ctx->ev1);
    for(hipEvent_t e : ctx->evs) hipEventDestroy(e);
    for(hipEvent_t e : ctx->evj) hipEventDestroy(e);
    for(hipEvent_t e : ctx->evp) hipEventDestroy(e);
    if(ctx->acc_stream) hipStreamDestroy(ctx->acc_stream);
    if(ctx->rk4b_stream) hipStreamDestroy(ctx->rk4b_stream);
    if(ctx->pp_stream) hipStreamDestroy(ctx->pp_stream);
    if(ctx->ev_cu) hipEventDestroy(ctx->ev_cu);
    if(ctx->rk4_cu_stream) hipStreamDestroy(ctx->rk4_cu_stream);
    if(ctx->pp_cu_stream) hipStreamDestroy(ctx->pp_cu_stream);
    if(ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return GEOAC_OK;
}

// A second context on the same device that shares the source's atmosphere tables (device memory, read-only to the kernels) and carries its
// parameters and launch-plan options: for callers that run several independent fans at once (the eigenray searches integrate the ray groups
// of a decision round concurrently, each group on its own context and streams).  Valid until the source uploads another atmosphere or is
// destroyed; destroy the clone first.
int geoac_clone(geoac_ctx* src, geoac_ctx** out){
    if(!src || !out) return GEOAC_E_INVALID;
    if(!src->have_atmo) return fail(src, GEOAC_E_INVALID, "clone: no atmosphere uploaded");
    geoac_ctx* c = nullptr;
    int rc = geoac_create(&c, src->eqset, src->device);
    if(rc) return rc;
    c->prm = src->prm; c->have_params = src->have_params; c->have_atmo = true; c->have_grid = src->have_grid;
    c->x = src->x; c->T = src->T; c->u = src->u; c->v = src->v; c->rho = src->rho; c->sl = src->sl; c->n_nodes = src->n_nodes;
    c->gnx = src->gnx; c->gny = src->gny; c->gx = src->gx; c->gy = src->gy; c->gtab_bytes = src->gtab_bytes;
    c->src_gen = src->atmo_gen; c->src_gen_at_clone = src->atmo_gen->load();
    c->seg.view_of(src->seg); c->rhot.view_of(src->rhot);
    c->d_gx.view_of(src->d_gx); c->d_gy.view_of(src->d_gy); c->d_gz.view_of(src->d_gz); c->d_gtab.view_of(src->d_gtab); c->d_gtab8.view_of(src->d_gtab8);
    if(src->have_grid){                                           // (k_init writes the absorption model's reference state here: one block per context)
        hipError_t e = c->d_consts.ensure(sizeof(double) * 8);
        if(e != hipSuccess){ geoac_destroy(c); return hipfail(src, e, "clone: constants block"); }
    }
    c->sort_rays = src->sort_rays; c->tile_rays = src->tile_rays; c->pp_onetrip = src->pp_onetrip; c->pp_lds_table = src->pp_lds_table; c->pp_lds_pad = src->pp_lds_pad; c->cu_split = src->cu_split; c->chunk_gib = src->chunk_gib; c->accum_batch = src->accum_batch; c->sub_test_stall = src->sub_test_stall; c->no_quad = src->no_quad; c->oct = src->oct; c->hex = src->hex; c->grid_lanes = src->grid_lanes; c->spread_override = src->spread_override;
    c->compact = src->compact; c->quad_cache = src->quad_cache; c->sub_min_waves = src->sub_min_waves; c->sub_epochs = src->sub_epochs; c->grid_coop = src->grid_coop;
    c->smp_cap = src->smp_cap; c->s_rows_override = src->s_rows_override; c->no_overlap = src->no_overlap; c->pp_blocks = src->pp_blocks; c->pair_frac = src->pair_frac; c->stagger_frac = src->stagger_frac; c->stagger_rows = src->stagger_rows;
    c->hybrid_rows = src->hybrid_rows; c->two_chunks = src->two_chunks; c->no_gate = src->no_gate; c->no_pair = src->no_pair; c->duo = src->duo; c->trio = src->trio; c->abs_table = src->abs_table;
    c->ev_slack = src->ev_slack; c->grid_build_host = src->grid_build_host; c->atab_tol = src->atab_tol; c->ppfix_cap = src->ppfix_cap;
    *out = c;
    return GEOAC_OK;
}

int geoac_set_stream(geoac_ctx* ctx, void* hip_stream){
    if(!ctx) return GEOAC_E_INVALID;
    if(ctx->own_stream && ctx->stream){ hipStreamSynchronize(ctx->stream); hipStreamDestroy(ctx->stream); }
    ctx->stream = (hipStream_t)hip_stream; ctx->own_stream = false;
    return GEOAC_OK;
}

int geoac_upload_atmo_1d(geoac_ctx* ctx, int n, const double* x, const double* T, const double* u,
                         const double* v, const double* rho, const double* slopes4){
    if(!ctx || n < 3 || !x || !T || !u || !v || !rho || !slopes4) return fail(ctx, GEOAC_E_INVALID, "upload_atmo_1d: bad arguments");
    if(ctx->eqset != GEOAC_EQ_2D && ctx->eqset != GEOAC_EQ_3D && ctx->eqset != GEOAC_EQ_GLOBAL)
        return fail(ctx, GEOAC_E_UNSUPPORTED, "1-D atmosphere on a range-dependent equation set");
    for(int i = 1; i < n; i++) if(!(x[i] > x[i - 1])) return fail(ctx, GEOAC_E_INVALID, "upload_atmo_1d: abscissa not strictly increasing");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->n_nodes = n;
    ctx->x.assign(x, x + n); ctx->T.assign(T, T + n); ctx->u.assign(u, u + n); ctx->v.assign(v, v + n); ctx->rho.assign(rho, rho + n);
    ctx->sl.assign(slopes4, slopes4 + 4 * (size_t)n);
    const double* sT = slopes4; const double* su = slopes4 + n; const double* sv = slopes4 + 2 * (size_t)n; const double* sr = slopes4 + 3 * (size_t)n;
    int nseg = n - 1;
    std::vector<double> seg((size_t)nseg * GEOAC_SEGW), rt((size_t)nseg * 4);
    for(int k = 0; k < nseg; k++){
        double* s = &seg[(size_t)k * GEOAC_SEGW];
        s[0] = x[k]; s[1] = x[k + 1];
        seg_coeffs(x[k], x[k + 1], T[k], T[k + 1], sT[k], sT[k + 1], s + 2, true);
        seg_coeffs(x[k], x[k + 1], u[k], u[k + 1], su[k], su[k + 1], s + 6, true);
        seg_coeffs(x[k], x[k + 1], v[k], v[k + 1], sv[k], sv[k + 1], s + 10, true);
        seg_coeffs(x[k], x[k + 1], rho[k], rho[k + 1], sr[k], sr[k + 1], &rt[(size_t)k * 4], false);
    }
    HIPCHK(ctx->seg.ensure(seg.size() * sizeof(double)));
    HIPCHK(ctx->rhot.ensure(rt.size() * sizeof(double)));
    HIPCHK(hipMemcpyAsync(ctx->seg.p, seg.data(), seg.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->rhot.p, rt.data(), rt.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // GeoAc_SetPropRegion (G2S_Spline1D.cpp:22-28 / G2S_GlobalSpline1D.cpp:22-30): vert_limit = top node
    if(!(ctx->prm.vert_limit == ctx->prm.vert_limit)) ctx->prm.vert_limit = x[n - 1];
    ctx->have_atmo = true;
    ctx->atmo_version++; ctx->atmo_gen->fetch_add(1);
    ctx->ran = false;                             // (the probes launch with the tables of the last fan: not after a new upload)
    return GEOAC_OK;
}

int geoac_upload_atmo_3d(geoac_ctx* ctx, int nx, int ny, int nz, const double* x, const double* y, const double* z,
                         const double* T, const double* u, const double* v, const double* rho){
    if(!ctx || nx < 2 || ny < 2 || nz < 3 || !x || !y || !z || !T || !u || !v || !rho) return fail(ctx, GEOAC_E_INVALID, "upload_atmo_3d: bad arguments");
    if(ctx->eqset != GEOAC_EQ_3D_RNGDEP && ctx->eqset != GEOAC_EQ_GLOBAL_RNGDEP) return fail(ctx, GEOAC_E_UNSUPPORTED, "grid atmosphere needs one of the range-dependent equation sets");
    for(int i = 1; i < nx; i++) if(!(x[i] > x[i - 1])) return fail(ctx, GEOAC_E_INVALID, "upload_atmo_3d: x not strictly increasing");
    for(int i = 1; i < ny; i++) if(!(y[i] > y[i - 1])) return fail(ctx, GEOAC_E_INVALID, "upload_atmo_3d: y not strictly increasing");
    for(int i = 1; i < nz; i++) if(!(z[i] > z[i - 1])) return fail(ctx, GEOAC_E_INVALID, "upload_atmo_3d: z not strictly increasing");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t tab_n = geoac_grid_table_size(nx, ny, nz), nn = (size_t)nx * ny;
    HIPCHK(ctx->d_gx.ensure(sizeof(double) * nx)); HIPCHK(ctx->d_gy.ensure(sizeof(double) * ny)); HIPCHK(ctx->d_gz.ensure(sizeof(double) * nz));
    HIPCHK(ctx->d_gtab.ensure(sizeof(double) * tab_n));
    HIPCHK(ctx->d_consts.ensure(sizeof(double) * 8));
    HIPCHK(hipMemcpyAsync(ctx->d_gx.p, x, sizeof(double) * nx, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_gy.p, y, sizeof(double) * ny, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_gz.p, z, sizeof(double) * nz, hipMemcpyHostToDevice, ctx->stream));
    if(ctx->grid_build_host){
        // the host builder (geoac_host.cpp), kept as the checker of the device builder and for A/B timing
        std::vector<double> tab(tab_n);
        geoac_grid_table_eq(ctx->eqset, nx, ny, nz, x, y, z, T, u, v, rho, tab.data());
        HIPCHK(hipMemcpyAsync(ctx->d_gtab.p, tab.data(), sizeof(double) * tab_n, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    } else {
        // device table (geoac_rngdep.h) built on the device (geoac_gridbuild.hip): 12 vertical spline systems per node + expansion;
        // only the raw fields cross PCIe (32 B per grid point instead of 1 120 B of table per point)
        DevBuf fields, work;
        HIPCHK(fields.ensure(sizeof(double) * 4 * nn * nz));
        hipError_t we = work.ensure(sizeof(double) * geoac_gridbuild_work_doubles(nx, ny, nz));
        if(we != hipSuccess){ fields.release(); return hipfail(ctx, we, "grid table scratch"); }
        const double* F[4] = { T, u, v, rho };
        hipError_t e = hipSuccess;
        for(int f = 0; f < 4 && e == hipSuccess; f++)
            e = hipMemcpyAsync((double*)fields.p + (size_t)f * nn * nz, F[f], sizeof(double) * nn * nz, hipMemcpyHostToDevice, ctx->stream);
        if(e == hipSuccess) e = geoac_gridbuild_launch(ctx->eqset == GEOAC_EQ_GLOBAL_RNGDEP ? 1 : 0, nx, ny, nz, (const double*)ctx->d_gx.p, (const double*)ctx->d_gy.p,
                                                       (const double*)ctx->d_gz.p, (const double*)fields.p, (double*)work.p, (double*)ctx->d_gtab.p, ctx->stream);
        if(e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        fields.release(); work.release();
        if(e != hipSuccess) return hipfail(ctx, e, "grid table build");
    }
    ctx->gtab_bytes = sizeof(double) * tab_n;
    if(ctx->eqset == GEOAC_EQ_3D_RNGDEP && geoac_kernels_cart_rec() == 32){
        // Cartesian kernels read 256-byte line-aligned records (eight cubics: V_x = D_x F, V_y = D_y F are not stored twice)
        ctx->gtab_bytes = sizeof(double) * geoac_gridpack_doubles(nx, ny, nz);
        HIPCHK(ctx->d_gtab8.ensure(ctx->gtab_bytes));
        HIPCHK(geoac_gridpack_launch(nx, ny, nz, (const double*)ctx->d_gtab.p, (double*)ctx->d_gtab8.p, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    } else ctx->d_gtab8.release();
    ctx->gnx = nx; ctx->gny = ny; ctx->n_nodes = nz;
    ctx->gx.assign(x, x + nx); ctx->gy.assign(y, y + ny); ctx->x.assign(z, z + nz);
    // GeoAc_SetPropRegion (G2S_MultiDimSpline3D.cpp:25-33)
    if(!(ctx->prm.vert_limit == ctx->prm.vert_limit)) ctx->prm.vert_limit = z[nz - 1];
    const double ext[4] = { x[0], x[nx - 1], y[0], y[ny - 1] };
    for(int q = 0; q < 4; q++) if(!(ctx->prm.xy_limits[q] == ctx->prm.xy_limits[q])) ctx->prm.xy_limits[q] = ext[q];
    ctx->have_grid = true; ctx->have_atmo = true;
    ctx->atmo_version++; ctx->atmo_gen->fetch_add(1);
    ctx->ran = false;                             // (the probes launch with the tables of the last fan: not after a new upload)
    return GEOAC_OK;
}

int geoac_grid_table_fetch(geoac_ctx* ctx, double* tab, size_t cap){
    if(!ctx || !tab) return GEOAC_E_INVALID;
    if(!ctx->have_grid) return fail(ctx, GEOAC_E_INVALID, "grid_table_fetch: no grid atmosphere uploaded");
    const size_t n = geoac_grid_table_size(ctx->gnx, ctx->gny, ctx->n_nodes);
    if(cap < n) return fail(ctx, GEOAC_E_CAPACITY, "grid_table_fetch: buffer too small");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpy(tab, ctx->d_gtab.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    return GEOAC_OK;
}

int geoac_get_params(geoac_ctx* ctx, geoac_params* p){
    if(!ctx || !p) return GEOAC_E_INVALID;
    *p = ctx->prm;
    return GEOAC_OK;
}
// c, u, v, rho of the 1-D atmosphere at abscissa x (host evaluation; set-up / reporting only)
int geoac_medium_1d(geoac_ctx* ctx, double x, double out[4]){
    if(!ctx || !out) return GEOAC_E_INVALID;
    if(!ctx->have_atmo || ctx->have_grid) return fail(ctx, GEOAC_E_INVALID, "medium_1d: no 1-D atmosphere uploaded");
    const size_t n = (size_t)ctx->n_nodes;
    out[0] = sqrt(kGamR * host_spline_f(ctx->x, ctx->T, ctx->sl.data(), x));
    out[1] = host_spline_f(ctx->x, ctx->u, ctx->sl.data() + n, x);
    out[2] = host_spline_f(ctx->x, ctx->v, ctx->sl.data() + 2 * n, x);
    out[3] = host_spline_f(ctx->x, ctx->rho, ctx->sl.data() + 3 * n, x);
    return GEOAC_OK;
}
int geoac_get_eqset(geoac_ctx* ctx, int* eqset){
    if(!ctx || !eqset) return GEOAC_E_INVALID;
    *eqset = ctx->eqset;
    return GEOAC_OK;
}

int geoac_set_params(geoac_ctx* ctx, const geoac_params* p){
    if(!ctx || !p) return GEOAC_E_INVALID;
    if(p->bounces < 0 || p->bounces + 1 > GEOAC_MAXLEGS) return fail(ctx, GEOAC_E_INVALID, "bounces out of range (0..63)");
    if(!(p->ds_min > 0) || !(p->ds_max >= p->ds_min)) return fail(ctx, GEOAC_E_INVALID, "ds_min/ds_max");
    ctx->prm = *p;
    if(!(ctx->prm.vert_limit == ctx->prm.vert_limit) && ctx->have_atmo) ctx->prm.vert_limit = ctx->x[ctx->n_nodes - 1];
    if(ctx->have_grid){
        const double ext[4] = { ctx->gx.front(), ctx->gx.back(), ctx->gy.front(), ctx->gy.back() };
        for(int q = 0; q < 4; q++) if(!(ctx->prm.xy_limits[q] == ctx->prm.xy_limits[q])) ctx->prm.xy_limits[q] = ext[q];
    }
    if(ctx->prm.mode & GEOAC_MODE_WRITE_CAUSTICS) ctx->prm.calc_amp = 1;      // GeoAcGlobal_main.cpp:166
    if(ctx->prm.sample_stride <= 0) ctx->prm.sample_stride = 25;
    ctx->have_params = true;
    return GEOAC_OK;
}

int geoac_fan_set_angles(geoac_ctx* ctx, int n_rays, const double* theta_deg, const double* phi_deg){
    if(!ctx || n_rays <= 0 || !theta_deg || !phi_deg) return fail(ctx, GEOAC_E_INVALID, "fan_set_angles: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->n_rays = n_rays;
    ctx->n_pad = (n_rays + 63) / 64 * 64;
    std::vector<double> ths, phs; std::vector<int> order;
    ctx->have_perm = false;
    const double* th_up = theta_deg; const double* ph_up = phi_deg; size_t n_up = (size_t)n_rays;
    if(ctx->sort_rays){
        std::vector<int> sorted((size_t)n_rays);
        for(int i = 0; i < n_rays; i++) sorted[(size_t)i] = i;
        const bool grid_set = (ctx->eqset == GEOAC_EQ_3D_RNGDEP || ctx->eqset == GEOAC_EQ_GLOBAL_RNGDEP);
        if(grid_set && ctx->tile_rays){
            // grid sets: a wave's gather costs what its lanes' DISTINCT (segment, cell) keys cost, and neighbours in BOTH launch angles stay in
            // one key longest: Z-order over (inclination rank, azimuth rank) - 64 consecutive slots are an 8 x 8 tile of the fan
            // (config-4 share: 44.7 -> 26.0 distinct keys per wave-stage, 1.325 -> 1.243 s; tools/order_probe.py)
            std::vector<double> ut(theta_deg, theta_deg + n_rays), up(phi_deg, phi_deg + n_rays);
            std::sort(ut.begin(), ut.end()); ut.erase(std::unique(ut.begin(), ut.end()), ut.end());
            std::sort(up.begin(), up.end()); up.erase(std::unique(up.begin(), up.end()), up.end());
            std::vector<unsigned long long> code((size_t)n_rays);
            auto spread_bits = [](unsigned long long v){ unsigned long long o = 0; for(int b = 0; b < 32; b++) o |= ((v >> b) & 1ull) << (2 * b); return o; };
            for(int i = 0; i < n_rays; i++){
                const unsigned long long rt = (unsigned long long)(std::lower_bound(ut.begin(), ut.end(), theta_deg[i]) - ut.begin());
                const unsigned long long rp = (unsigned long long)(std::lower_bound(up.begin(), up.end(), phi_deg[i]) - up.begin());
                code[(size_t)i] = (spread_bits(rt) << 1) | spread_bits(rp);
            }
            std::stable_sort(sorted.begin(), sorted.end(), [&](int a, int b){ return code[(size_t)a] < code[(size_t)b]; });
        } else {
            // stable sort of the rays by launch inclination, as an LSD radix sort on the order-preserving integer image of the doubles (four 16-bit digits): the same
            // permutation std::stable_sort with `theta[a] < theta[b]` gives, in 0.15 ms instead of 1 ms for the metric fan's 32 400 rays - inside the timed region
            auto key_of = [](double v){ uint64_t u; memcpy(&u, &v, 8); return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull); };
            bool plain = true;                                        // (NaN or -0.0 / +0.0 mixtures would order differently than `<`: leave those to the comparison sort)
            std::vector<uint64_t> key((size_t)n_rays);
            for(int i = 0; i < n_rays; i++){
                const double v = theta_deg[i];
                if(!(v == v) || (v == 0.0 && std::signbit(v))) plain = false;
                key[(size_t)i] = key_of(v);
            }
            if(plain){
                std::vector<int> tmp((size_t)n_rays);
                std::vector<uint32_t> cnt(65536);
                for(int pass = 0; pass < 4; pass++){
                    const int sh = 16 * pass;
                    std::fill(cnt.begin(), cnt.end(), 0u);
                    for(int i = 0; i < n_rays; i++) cnt[(key[(size_t)sorted[(size_t)i]] >> sh) & 0xffff]++;
                    uint32_t run = 0;
                    for(auto& c : cnt){ const uint32_t n = c; c = run; run += n; }
                    for(int i = 0; i < n_rays; i++){ const int r = sorted[(size_t)i]; tmp[cnt[(key[(size_t)r] >> sh) & 0xffff]++] = r; }
                    sorted.swap(tmp);
                }
            } else
            std::stable_sort(sorted.begin(), sorted.end(), [&](int a, int b){ return theta_deg[a] < theta_deg[b]; });
        }
        // slot -> ray (-1: a slot without a ray: the tail padding of the last wave)
        order.assign(sorted.begin(), sorted.end());
        ths.resize((size_t)n_rays); phs.resize((size_t)n_rays);
        for(int i = 0; i < n_rays; i++){ ths[(size_t)i] = theta_deg[sorted[(size_t)i]]; phs[(size_t)i] = phi_deg[sorted[(size_t)i]]; }
        ctx->n_pad = (int)((order.size() + 63) / 64 * 64);
        while(order.size() < (size_t)ctx->n_pad) order.push_back(-1);
        th_up = ths.data(); ph_up = phs.data(); n_up = ths.size();
        HIPCHK(ctx->perm.ensure(sizeof(int) * (size_t)ctx->n_pad));
        HIPCHK(hipMemcpyAsync(ctx->perm.p, order.data(), sizeof(int) * (size_t)ctx->n_pad, hipMemcpyHostToDevice, ctx->stream));
        ctx->have_perm = true;
    }
    HIPCHK(ctx->theta.ensure(sizeof(double) * (size_t)ctx->n_pad));
    HIPCHK(ctx->phi.ensure(sizeof(double) * (size_t)ctx->n_pad));
    HIPCHK(hipMemcpyAsync(ctx->theta.p, th_up, sizeof(double) * n_up, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->phi.p, ph_up, sizeof(double) * n_up, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->have_angles = true;
    return GEOAC_OK;
}

// one attempt at the fan under the context's current launch plan.  GEOAC_RETRY: the attempt found that a plan feature does not hold on this
// device / input (sub-epoch hand-off, absorption table), switched it off for this context and wants the fan run again (geoac_fan_launch)
static const int GEOAC_RETRY = -1000;
static int fan_launch_once(geoac_ctx* ctx){
    if(!ctx->have_atmo || !ctx->have_angles) return fail(ctx, GEOAC_E_INVALID, "fan_launch: atmosphere and angles must be uploaded first");
    const bool is_grid = (ctx->eqset == GEOAC_EQ_3D_RNGDEP || ctx->eqset == GEOAC_EQ_GLOBAL_RNGDEP);
    if(is_grid != ctx->have_grid) return fail(ctx, GEOAC_E_INVALID, "fan_launch: atmosphere kind does not match the equation set");
    HIPCHK(hipSetDevice(ctx->device));
    const geoac_params& p = ctx->prm;
    GeoacDevParams P{};
    P.eqset = ctx->eqset; P.calc_amp = p.calc_amp ? 1 : 0; P.mode = p.mode; P.bounces = p.bounces;
    P.n_rays = ctx->n_rays; P.n_pad = ctx->n_pad;
    const bool is_global = (ctx->eqset == GEOAC_EQ_GLOBAL);                                  // stratified spherical set (pair kernel, range test)
    const bool is_sph = is_global || ctx->eqset == GEOAC_EQ_GLOBAL_RNGDEP;                   // geocentric radius as the height coordinate
    if(is_global){ P.E = p.calc_amp ? 18 : 6; P.pathw = 6; }
    else if(ctx->eqset == GEOAC_EQ_3D){ P.E = p.calc_amp ? 12 : 4; P.pathw = 4; }
    else if(is_grid){ P.E = p.calc_amp ? 18 : 6; P.pathw = 6; }
    else { P.E = p.calc_amp ? 6 : 3; P.pathw = 2; }
    P.rays_form = ((p.mode & (GEOAC_MODE_WRITE_RAYS | GEOAC_MODE_WRITE_CAUSTICS)) != 0 || ctx->eqset == GEOAC_EQ_2D) ? 1 : 0;
    P.nseg = ctx->n_nodes - 1;
    P.step_limit = (long long)(p.ray_limit * (int)(1.0 / (p.ds_min * 10)));   // GeoAc.Solver.cpp:14
    if(P.step_limit > 0x7fffffffLL) P.step_limit = 0x7fffffffLL;             // (the kernels count a leg's steps in 32 bits; the reference's own `int step_limit` overflows beyond this)
    if(P.step_limit < 2) P.step_limit = 2;
    P.x_min = ctx->x[0]; P.x_max = ctx->x[ctx->n_nodes - 1];
    P.ds_min = p.ds_min; P.ds_max = p.ds_max;
    P.r_earth = is_sph ? p.r_earth : 0.0; P.z_grnd = p.z_grnd;
    P.ground = P.r_earth + p.z_grnd;
    P.vert_limit = p.vert_limit; P.range_limit = p.range_limit;
    if(p.range_limit >= 0.0){ const double l2 = p.range_limit * p.range_limit; P.range_sq[0] = l2 * (1.0 - 1e-12); P.range_sq[1] = l2 * (1.0 + 1e-12); }
    else { P.range_sq[0] = 0.0; P.range_sq[1] = -1.0; }              // (a negative limit: every row is beyond it, as r > limit says)
    if(is_global){
        double half = p.range_limit / (2.0 * p.r_earth);
        if(half >= kPi / 2.0){ P.range_thresh = 2.0; P.range_skip = 1e300; }      // asin saturates: the range test can never fire
        else if(half <= 0.0){ P.range_thresh = -1.0; P.range_skip = -1.0; }       // (always fires)
        else { double s = sin(half); P.range_thresh = s * s; P.range_skip = half * (1.0 - 1e-9); }
    }
    P.src[0] = p.src[0]; P.src[1] = p.src[1]; P.src[2] = p.src[2];
    P.freq = p.freq; P.tweak_abs = p.tweak_abs;
    {   // SuthBass reference state: T_o, P_o at abscissa z_grnd (Atmo_State.Absorption.Global.cpp:31-32 passes the km
        // altitude as a radius, which clamps to the lowest node; the Cartesian twin evaluates at z = z_grnd)
        P.sb_const[0] = pow(10.0, -0.67887); P.sb_const[1] = pow(10.0, -0.10744); P.sb_const[2] = pow(10, -3.3979);
        P.sb_const[3] = 5.0 / sqrt(21.0); P.sb_const[4] = sqrt(3.0 / 7.0);
        if(is_grid){ P.T_o = P.P_o = 0.0; P.c000 = 0.0; } else {
        double Tg = host_spline_f(ctx->x, ctx->T, ctx->sl.data(), p.z_grnd);
        double rg = host_spline_f(ctx->x, ctx->rho, ctx->sl.data() + 3 * (size_t)ctx->n_nodes, p.z_grnd);
        double cg = sqrt(kGamR * Tg) * 1000.0;
        P.T_o = cg * cg / (kRgas * kGam);
        P.cbrt_To = cbrt(P.T_o);
        P.P_o = rg * (cg * cg) / kGam * 1000.0;
        P.c000 = sqrt(kGamR * host_spline_f(ctx->x, ctx->T, ctx->sl.data(), 0.0));       // c(0,0,0), 3DStratified.cpp:367
        }
        P.src_trig[0] = sin(p.src[1] * kPi / 180.0); P.src_trig[1] = cos(p.src[1] * kPi / 180.0);
    }
    // ---- epoch size: 8192 rows unless a path chunk would pass 40 GiB (three chunks + three contrib buffers <= 160 GiB of the 288).
    //      Measured on the metric fan (GEOAC_S_ROWS sweep, hybrid build): 4096 rows 179 ms per pass, 8192 167, 12288 167, 16384 173 -
    //      every epoch boundary costs a launch + table reload, while very long epochs leave the last post-pass uncovered ----
    size_t row_bytes = (size_t)P.pathw * P.n_pad * sizeof(double);
    // (CHUNK_GIB: the cap per path chunk; a device that is shared - several ranks or contexts, torch tensors - can be given a smaller one than the 40 GiB that suit a GPU of its own)
    long long s_rows = (long long)(((unsigned long long)ctx->chunk_gib << 30) / row_bytes);      // round 2: 40 GiB per chunk (config 3: 2760 -> 6900 rows, 21 -> 9 epochs, +8 %); the free-memory clamp below still applies
    if(s_rows > 8192) s_rows = 8192;
    if(s_rows < 64) s_rows = 64;
    if(ctx->s_rows_override >= 8) s_rows = ctx->s_rows_override;
    {   // the chunks must fit in what the device has left (another context, torch tensors, RCCL buffers may share it): shorter epochs
        // rather than an out-of-memory failure.  Budget: 80 % of the free memory plus what this context's chunks already hold.
        size_t free_b = 0, total_b = 0;
        if(hipMemGetInfo(&free_b, &total_b) == hipSuccess){
            size_t held = 0;
            for(int b = 0; b < 3; b++) held += ctx->path[b].bytes + ctx->contrib[b].bytes;
            const size_t per_row = (size_t)(ctx->two_chunks ? 2 : 3) * (row_bytes + 2 * (size_t)P.n_pad * sizeof(double));
            const double budget = 0.8 * (double)(free_b + held);
            if((double)per_row * (double)s_rows > budget){
                long long fit = (long long)(budget / (double)per_row);
                if(fit < 16) return fail(ctx, GEOAC_E_NOMEM, "fan_launch: not enough free device memory for the path chunks of this fan (" + std::to_string(free_b >> 20) + " MiB free)");
                s_rows = fit;
            }
        }
    }
    P.s_rows = (int)s_rows;
    size_t lds_need = (size_t)P.nseg * GEOAC_SEGW * sizeof(double);
    P.table_in_lds = (!is_grid && lds_need <= 160 * 1024) ? 1 : 0;
    if(is_grid){
        P.gnx = ctx->gnx; P.gny = ctx->gny;
        P.g_lo[0] = ctx->gx.front(); P.g_hi[0] = ctx->gx.back(); P.g_lo[1] = ctx->gy.front(); P.g_hi[1] = ctx->gy.back();
        P.gx = (const double*)ctx->d_gx.p; P.gy = (const double*)ctx->d_gy.p; P.gz = (const double*)ctx->d_gz.p;
        P.gtab = (const double*)(ctx->d_gtab8.p ? ctx->d_gtab8.p : ctx->d_gtab.p); P.dev_consts = (double*)ctx->d_consts.p;
        for(int q = 0; q < 4; q++) P.xy_lim[q] = p.xy_limits[q];
    }
    ctx->legs = p.bounces + 1;

    HIPCHK(ctx->state.ensure(sizeof(double) * (size_t)ST_NSTATE * P.n_pad));
    const bool sampling = (p.mode & (GEOAC_MODE_WRITE_RAYS | GEOAC_MODE_WRITE_CAUSTICS)) != 0;
    P.smp_stride = p.sample_stride > 0 ? p.sample_stride : 25;
    P.ev_cap = sampling ? (P.s_rows / P.smp_stride + ctx->ev_slack) : 0;
    P.smp_cap = sampling ? ctx->smp_cap : 0;
    if(sampling){
        HIPCHK(ctx->smp_out.ensure(sizeof(double) * GEOAC_SMP_STRIDE * (size_t)P.smp_cap));
        P.smp_out = (double*)ctx->smp_out.p;
    }
    HIPCHK(ctx->rec.ensure(sizeof(double) * (size_t)ctx->n_rays * ctx->legs * GEOAC_REC_STRIDE));
    HIPCHK(ctx->counters.ensure(32 * sizeof(unsigned long long)));
    P.seg = (const double*)ctx->seg.p; P.rho = (const double*)ctx->rhot.p;
    P.theta_deg = (const double*)ctx->theta.p; P.phi_deg = (const double*)ctx->phi.p;
    P.perm = ctx->have_perm ? (const int*)ctx->perm.p : nullptr;
    P.state = (double*)ctx->state.p;
    P.rec = (double*)ctx->rec.p; P.counters = (unsigned long long*)ctx->counters.p;
    // absorption table (stratified sets; k_atab_build): alpha is a function of the height coordinate alone there
    P.atab = nullptr; P.lat_trig = nullptr; P.atab_on = 0; P.atab_D = 0.0; P.atab_lo = 0.0; P.ppfix = nullptr; P.ppfix_cap = 0;
    P.seg_per_x = (double)P.nseg / (P.x_max - P.x_min);
    if(!is_grid && ctx->abs_table){
        const double D = std::max(std::min(0.05, p.ds_max), p.ds_min), tol = ctx->atab_tol;
        const double key[7] = { P.freq, P.tweak_abs, P.T_o, P.P_o, P.r_earth, D, tol };
        P.atab_D = D; P.atab_lo = P.x_min - D;
        if(ctx->atab_version != ctx->atmo_version || memcmp(key, ctx->atab_key, sizeof(key)) != 0){
            const size_t n_ent = (size_t)P.nseg + 2;
            HIPCHK(ctx->atab.ensure(sizeof(double) * (GEOAC_ATABW * n_ent + 2 * (size_t)GEOAC_LAT_N)));      // (+ the latitude table behind it)
            HIPCHK(geoac_launch_atab_build(&P, (double*)ctx->atab.p, tol, ctx->stream));
            std::vector<double> h(GEOAC_ATABW * n_ent);
            HIPCHK(hipMemcpyAsync(h.data(), ctx->atab.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            ctx->atab_entries = (int)n_ent; ctx->atab_flagged = 0; ctx->atab_worst = 0.0;
            for(size_t e = 0; e < n_ent; e++){
                if(h[e * GEOAC_ATABW] < 0.0) ctx->atab_flagged++;
                else ctx->atab_worst = std::max(ctx->atab_worst, h[e * GEOAC_ATABW + GEOAC_ATABW - 1]);
            }
            memcpy(ctx->atab_key, key, sizeof(key));
            ctx->atab_version = ctx->atmo_version;
        }
        // a profile most of whose segments the interpolant cannot serve (very long segments) keeps the exact post-pass
        if(4 * ctx->atab_flagged <= ctx->atab_entries){
            P.atab = (const double*)ctx->atab.p; P.atab_on = 1;
            P.lat_trig = P.atab + (size_t)GEOAC_ATABW * ((size_t)P.nseg + 2);
            P.ppfix_cap = ctx->ppfix_cap;
            HIPCHK(ctx->ppfix.ensure(sizeof(int) * 2 * (size_t)P.ppfix_cap));
            P.ppfix = (int*)ctx->ppfix.p;
        }
    }

    // two lanes per ray (EqGlobalPair) for the Global set with amplitudes when no sample capture is requested
    {   // largest displacement of one RK4 step: ds = max(min(0.05 - ..., ds_max), ds_min)  (GeoAc_Set_ds)
        double ds_bound = std::max(std::min(0.05, p.ds_max), p.ds_min);
        double hmin = 1e300;
        for(int i = 1; i < ctx->n_nodes; i++) hmin = std::min(hmin, ctx->x[i] - ctx->x[i - 1]);
        P.seg_safe = (hmin >= 1.001 * ds_bound) ? 1 : 0;
    }
    P.pp_blocks = ctx->pp_blocks;
    P.slot_lo = 0; P.slot_hi = P.n_pad; P.live_slot = 1;
    // two lanes per ray shorten the serial chain (x1.25) at twice the lanes: only worth it while the fan leaves SIMDs idle
    P.lanes_per_ray = ((is_global || ctx->eqset == GEOAC_EQ_3D) && p.calc_amp && !sampling && !ctx->no_pair && (long long)P.n_pad * 2 / 64 <= 1024) ? 2 : 1;
    // Global set with amplitudes, arrivals only, a fan of at most one workgroup per CU (128 rays each) and a profile whose packed table
    // leaves room for the message slots: the wave-specialised kernel (geoac_duo.h) - the ray on one wave, its two launch-angle derivative
    // systems on a second one
    P.duo = (is_global && p.calc_amp && !sampling && ctx->duo && !ctx->no_pair && P.table_in_lds && geoac_duo_lds(P.nseg) <= 160 * 1024 &&
             (long long)P.n_pad <= 256ll * 128) ? ctx->duo : 0;
    if(P.duo) P.lanes_per_ray = 1;
    // ... or the three-wave one (geoac_trio.h) for the slots the plan gives two lanes per ray: 64 rays per workgroup
    P.trio = (!P.duo && is_global && p.calc_amp && !sampling && ctx->trio && P.lanes_per_ray == 2 && P.table_in_lds && geoac_trio_lds(P.nseg) <= 160 * 1024) ? ctx->trio : 0;
    // grid sets, small fans: four lanes per ray (one cell corner each) while that still leaves one wave per SIMD
    // (16 385 - 32 768 rays used to take two lanes per ray: measured on MI355X, profiles/r03_midfans.txt, the cooperative one-lane kernel is faster
    // there for the Cartesian set - 4.8e8 against 4.1e8 ray-steps/s at 24 000 rays - and within 5 % for the spherical one, and it uses no scratch)
    if(is_grid && !ctx->no_quad){
        if((long long)P.n_pad * 4 / 64 <= 1024) P.lanes_per_ray = 4;
    }
    if(is_grid && ctx->grid_lanes) P.lanes_per_ray = ctx->grid_lanes;
    // grid sets with amplitudes, fans of a few rays (the eigenray rounds): eight lanes per ray - four cell corners x the two launch-angle
    // systems (EqGlobalRngDepOct, Eq3DRngDepOct) - while that is at most one wave per CU
    const bool oct_ok = is_grid && p.calc_amp && !sampling && ctx->quad_cache && (long long)P.n_pad * 8 / 64 <= 256;
    if(P.lanes_per_ray == 8 && !oct_ok) P.lanes_per_ray = 4;
    if(is_grid && !ctx->grid_lanes && !ctx->no_quad && ctx->oct && oct_ok && P.lanes_per_ray == 4) P.lanes_per_ray = 8;
    // ... and sixteen - four corners x (three fields + one) x ... the two systems on the halves - for the spherical set while THAT is at most one wave per CU (EqGlobalRngDepHex)
    const bool hex_ok = oct_ok && ctx->eqset == GEOAC_EQ_GLOBAL_RNGDEP && (long long)P.n_pad * 16 / 64 <= 256;
    if(P.lanes_per_ray == 16 && p.calc_amp && !hex_ok) P.lanes_per_ray = oct_ok ? 8 : 4;
    if(P.lanes_per_ray == 8 && !ctx->grid_lanes && ctx->hex && hex_ok) P.lanes_per_ray = 16;
    // ... and amplitude-less fans of the spherical set that leave room for sixteen lanes on the cached kernel at two waves per CU (EqGlobalRngDepScan16: the
    // inclination scans of a search with few receivers; cache_fits below has the last word)
    const bool scan16_ok = is_grid && ctx->eqset == GEOAC_EQ_GLOBAL_RNGDEP && !p.calc_amp && !sampling && ctx->quad_cache && (long long)P.n_pad * 16 / 64 <= 512;
    if(P.lanes_per_ray == 16 && !p.calc_amp && !scan16_ok) P.lanes_per_ray = 4;
    if(P.lanes_per_ray == 4 && !ctx->grid_lanes && !ctx->no_quad && ctx->hex && scan16_ok) P.lanes_per_ray = 16;
    // RK4 workgroup shape: with the table in LDS one workgroup owns a CU, so spread the waves over the 256 CUs
    int waves = P.n_pad * P.lanes_per_ray / 64;
    int wpb = (waves + 255) / 256;
    if(wpb < 1) wpb = 1;
    if(wpb > 4) wpb = 4;                       // k_rk4 is compiled with __launch_bounds__(256): never launch a larger workgroup
    int block = 64 * wpb;
    if(!P.table_in_lds) block = 64;
    // grid sets are bound by divergent table gathers (one cache line per active lane and load): while the fan has fewer waves than
    // the chip has SIMDs, thin the waves out (1 wave per SIMD is what the kernel's register budget allows)
    P.spread = 1;
    const bool coop_able = is_grid && ctx->grid_coop && ctx->gtab_bytes < (4ull << 30);
    if(is_grid && P.lanes_per_ray == 1){
        // measured (tools/perf_rngdep.py, GEOAC_SPREAD sweep): 2-4 way thinning gains 5-20 %, 8+ loses again (path stores and the
        // post-pass reads scatter), more waves than SIMDs loses a lot.  Only for the per-lane-gather kernel (GRID_COOP=0): a fan that
        // comes here on the default plan has more than 16 384 rays and takes the cooperative kernel unthinned (24 000-ray fan on the
        // 5 x 5 x 1400 grid: 5.1e8 ray-steps/s against 3.9e8 thinned two-way on the per-lane kernel, profiles/r03_c_midfans.txt)
        while(!coop_able && P.spread < 4 && (long long)P.n_pad * (P.spread * 2) / 64 <= 1024) P.spread *= 2;
        if(ctx->spread_override > 0){ P.spread = 1; while(P.spread * 2 <= ctx->spread_override && P.spread < 64) P.spread *= 2; }
    }
    // small multi-lane fans: records and z nodes cached in LDS (one wave per workgroup), the stage latency is what such a fan costs.  A wave needs its
    // step rows (2 x E x 512 B), 64 record slots of 976 B and the z nodes: one wave per CU with amplitudes (E = 18, or 12 per lane in the eight- and
    // sixteen-lane kernels), TWO without (E = 6: 80 KB each on a 1400-segment grid) - the inclination scans of an eigenray search (4 272 rays x 4
    // lanes = 267 waves for the 8-receiver share of config 5) stay on the cached kernel that way
    auto cache_fits = [&](int lanes){
        const int e_rows = !p.calc_amp ? 6 : ((lanes == 8 || lanes == 16) ? 12 : 18);
        const size_t per_wave = 2 * (size_t)e_rows * 64 * sizeof(double) + 64 * 976 + (size_t)(P.nseg + 1) * sizeof(double);
        if(!(is_grid && (lanes == 4 || lanes == 8 || lanes == 16) && ctx->quad_cache) || per_wave > 160 * 1024) return false;
        const long long waves_per_cu = (long long)((160 * 1024) / per_wave);
        return (long long)P.n_pad * lanes / 64 <= 256 * waves_per_cu;
    };
    P.quad_cache = cache_fits(P.lanes_per_ray) ? 1 : 0;
    if((P.lanes_per_ray == 8 || P.lanes_per_ray == 16) && !P.quad_cache){ P.lanes_per_ray = 4; P.quad_cache = cache_fits(4) ? 1 : 0; }      // (the eight- and sixteen-lane kernels exist with the record cache only)
    // dense one-lane-per-ray grid fans (more waves than SIMDs): the quads of a wave fetch the table records together (grid_eval3_coop)
    P.coop = (is_grid && P.lanes_per_ray == 1 && P.spread == 1 && ctx->grid_coop && ctx->gtab_bytes < (4ull << 30)) ? 1 : 0;   // (32-bit record offsets)

    // ---- hybrid fan (Global set, CalcAmp, inclination-sorted): the two-lane kernel shortens the serial chain of a ray by x1.25 but
    //      doubles its lanes, and with one wave on every SIMD the post-pass (168 VGPRs beside 384) cannot run next to the RK4 waves at
    //      all: the epochs then serialise RK4 and post-pass.  Only the longest rays set the finish time, and those are the shallow
    //      ones (metric fan: 54 k steps at 0.5 deg, <= 37 k above 5 deg): so the lowest-inclination share of the sorted fan gets two
    //      lanes, the rest one lane (a concurrent launch on a second stream with proportionally fewer rows per epoch so that both
    //      finish an epoch together), and the CUs that stay free run the post-pass throughout. ----
    int n_pair = 0;                                   // slots [0, n_pair): two lanes per ray; [n_pair, n_pad): one lane
    // launch-plan defaults, measured in one process per set (tools/sweep_hybrid.py): the spherical set's long rays are its shallow tenth
    // (0.10 / 0.75: 121.8 ms; 0.15: 124.7, 0.20: 126.9); the 3-D set's are spread over the inclinations (0.25 / 0.80: 108 ms; 0.10 / 0.75: 125)
    const double plan_pair_frac = ctx->pair_frac >= 0.0 ? ctx->pair_frac : (P.eqset == GEOAC_EQ_3D ? 0.25 : 0.10);
    const double plan_hybrid_rows = ctx->hybrid_rows > 0.0 ? ctx->hybrid_rows : (P.eqset == GEOAC_EQ_3D ? 0.80 : 0.75);
    const bool hybrid = (P.lanes_per_ray == 2 && !is_grid && ctx->have_perm && plan_pair_frac < 1.0 && !ctx->no_overlap &&
                         (long long)P.n_pad * 2 / 64 > 512);   // a fan that leaves half the SIMDs idle anyway keeps two lanes for every ray
    if(hybrid){
        n_pair = (int)(((long long)(plan_pair_frac * ctx->n_rays) + 127) / 128 * 128);
        if(n_pair >= P.n_pad) n_pair = P.n_pad;
    }
    bool split = hybrid && n_pair > 0 && n_pair < P.n_pad;      // n_pair == 0 (GEOAC_PAIR_FRAC=0): everything on the one-lane kernel, one launch
    if(hybrid && n_pair == 0) P.lanes_per_ray = 1;
    // live-ray compaction between epochs (single-launch fans; a hybrid fan assigns its two kernels by slot range): epoch e > 0 runs
    // over the dense list of the rays alive after epoch e-1, built on the device (k_compact) right before its RK4 launch
    const bool compact = ctx->compact && !hybrid;
    // ---- staggered epochs (stratified sets, fans with more waves than the chip has wave slots): every live ray advances one epoch per launch, a launch takes
    //      ceil(waves / 1024) rounds - so a ray that is twice as long as the average is still on its way when the chip has long run empty (config 3: 2 025 waves, rays of
    //      34 000 steps on average and 55 000 at most - the shallowest: a tail of seven epochs with a few dozen waves, 12 % of the RK4 time).  The launch is therefore cut in
    //      two in COLUMN space (the compacted list keeps the inclination order, the shallow rays come first): the leading share gets the epoch's rows, the others
    //      stagger_rows of them on a second stream - per unit of time the long rays advance further, and all of them end together.  Which columns got how many rows
    //      changes nothing in the records (test_full_fan_is_schedule_independent). ----
    //      Measured (tools/sweep_cu_split.py, profiles/r04_f_stagger.txt): config 3 354-362 -> 326-336 ms per pass, flat over shares of 0.10-0.30 and row ratios of 0.5-0.7;
    //      a 72 000-ray fan 253 -> 213 ms, a 200 000-ray one 415 -> 376, GeoAc3D 720 x 180 312-331 -> 297-303.
    const double plan_stagger_frac = ctx->stagger_frac >= 0.0 ? ctx->stagger_frac : 0.12;
    const double plan_stagger_rows = ctx->stagger_rows > 0.0 ? ctx->stagger_rows : 0.6;
    bool stagger = compact && !is_grid && !sampling && P.table_in_lds && ctx->have_perm && !ctx->no_overlap && plan_stagger_frac > 0.0 && P.lanes_per_ray == 1 && (long long)P.n_pad / 64 > 1024;   // (arrivals-only fans: what the plan was measured on)
    const bool stagger_plan = stagger;
    // fans that start with one lane per ray because two would not fit the chip (more than 1 024 two-lane waves): once the rays still alive do fit, the remaining epochs - the
    // serial chain of the few longest rays - run on the two-lane kernel (the same state layout; what a hybrid fan does when its two-lane share has finished)
    const bool late_pair_able = compact && !is_grid && (is_global || ctx->eqset == GEOAC_EQ_3D) && p.calc_amp && !sampling && !ctx->no_pair && P.table_in_lds && P.lanes_per_ray == 1 && !P.duo;
    bool late_pair = false;
    unsigned long long long_bound = (unsigned long long)(plan_stagger_frac * ctx->n_rays);      // live rays of the leading share (an upper bound: what its last launch counted)
    if(compact){
        for(int b = 0; b < 3; b++) HIPCHK(ctx->colmap[b].ensure(sizeof(int) * (size_t)P.n_pad));
        HIPCHK(ctx->ncols.ensure(4 * sizeof(int)));
    }
    P.n_cols_bound = P.n_pad;
    P.sub = 1; P.sub_w = 0; P.sub_flags = nullptr;
    const int n_chunks = ctx->two_chunks ? 2 : 3;     // three path chunks in rotation: the post-pass may lag the RK4 by more than one epoch (measured: GeoAc3D 360 x 90 fan 212 -> 160 ms; GEOAC_TWO_CHUNKS=1 for A/B)
    for(int b = 0; b < n_chunks; b++){
        if(sampling){                                             // per-chunk event lists of the WriteRays / WriteCaustics rows
            HIPCHK(ctx->ev_row[b].ensure(sizeof(int) * (size_t)P.ev_cap * P.n_pad));
            HIPCHK(ctx->ev_m[b].ensure(sizeof(int) * (size_t)P.ev_cap * P.n_pad));
            HIPCHK(ctx->ev_amp[b].ensure(sizeof(double) * (size_t)P.ev_cap * P.n_pad));
            HIPCHK(ctx->nev[b].ensure(sizeof(int) * (size_t)P.n_pad));
        }
        HIPCHK(ctx->nrows[b].ensure(sizeof(int) * (size_t)P.n_pad));
        HIPCHK(ctx->nlegend[b].ensure(sizeof(int) * (size_t)P.n_pad));
        HIPCHK(ctx->legend[b].ensure(sizeof(int) * (size_t)P.n_pad * GEOAC_MAXLEGS));
    }

    // the path chunks themselves.  The free-memory figure above can be stale by the time they are allocated (another process on the same
    // device sized its own chunks from the same figure): on an out-of-memory answer give the chunks back, halve the epoch and try again.
    for(;;){
        hipError_t e = hipSuccess;
        for(int b = 0; b < n_chunks && e == hipSuccess; b++){
            e = ctx->path[b].ensure(row_bytes * (size_t)P.s_rows);
            if(e == hipSuccess) e = ctx->contrib[b].ensure(sizeof(double) * 2 * (size_t)P.n_pad * P.s_rows);
        }
        if(e == hipSuccess) break;
        (void)hipGetLastError();
        for(int b = 0; b < 3; b++){ ctx->path[b].release(); ctx->contrib[b].release(); }
        if(e != hipErrorOutOfMemory || P.s_rows < 32 || ctx->s_rows_override >= 8) return hipfail(ctx, e, "path chunks");
        P.s_rows /= 2;
    }

    // ---- epoch pipeline: RK4 of epoch e on the context's stream, post-pass of epoch e on a second stream, chunk
    //      buffers alternate, so k_rk4(e+1) runs beside k_postpass(e)/k_accum(e).  The host only waits for the
    //      live-ray count of each RK4 launch (it must know when to stop). ----
    // CU partition (CU_SPLIT): fans whose RK4 waves and post-pass cannot share a SIMD (442 + 146 registers) and that fill the chip alternate between the two otherwise - an
    // RK4 workgroup needs a whole CU (the table in LDS), so a round of them waits until post-pass workgroups have drained from enough CUs, and the post-pass gets what
    // a round leaves over.  With two streams confined to disjoint sets of CUs (hipExtStreamCreateWithCUMask; the mask's bits go round robin over the XCDs, so either set is
    // spread evenly over the eight of them) both run all the time, each at the occupancy that suits it.
    int cu_split = ctx->cu_split >= 0 ? ctx->cu_split : 0;
    if(ctx->no_overlap || hybrid || !P.table_in_lds || is_grid) cu_split = 0;
    if(cu_split > 0){
        int n_cu = 0;
        HIPCHK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device));
        cu_split = std::min(std::max(cu_split / 8 * 8, 8), n_cu - 8);
        if(n_cu < 32 || n_cu > 256) cu_split = 0;
        else if(ctx->cu_split_made != cu_split){
            if(ctx->rk4_cu_stream){ hipStreamSynchronize(ctx->rk4_cu_stream); hipStreamDestroy(ctx->rk4_cu_stream); ctx->rk4_cu_stream = nullptr; }
            if(ctx->pp_cu_stream){ hipStreamSynchronize(ctx->pp_cu_stream); hipStreamDestroy(ctx->pp_cu_stream); ctx->pp_cu_stream = nullptr; }
            ctx->cu_split_made = 0;
            uint32_t m_pp[8] = {0}, m_rk[8] = {0};
            for(int i = 0; i < n_cu; i++) (i < cu_split ? m_pp : m_rk)[i / 32] |= 1u << (i % 32);
            if(hipExtStreamCreateWithCUMask(&ctx->rk4_cu_stream, 8, m_rk) != hipSuccess || hipExtStreamCreateWithCUMask(&ctx->pp_cu_stream, 8, m_pp) != hipSuccess){
                (void)hipGetLastError();                          // (a device / driver without CU masks: no partition)
                if(ctx->rk4_cu_stream){ hipStreamDestroy(ctx->rk4_cu_stream); ctx->rk4_cu_stream = nullptr; }
                ctx->pp_cu_stream = nullptr; cu_split = 0;
            } else ctx->cu_split_made = cu_split;
        }
    }
    hipStream_t s = cu_split ? ctx->rk4_cu_stream : ctx->stream, sp = ctx->no_overlap ? ctx->stream : (cu_split ? ctx->pp_cu_stream : ctx->pp_stream), sa = ctx->no_overlap ? ctx->stream : ctx->acc_stream;
    if(cu_split){                                                 // (whatever the caller queued on the context's stream comes first)
        if(!ctx->ev_cu) HIPCHK(hipEventCreateWithFlags(&ctx->ev_cu, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ctx->ev_cu, ctx->stream));
        HIPCHK(hipStreamWaitEvent(s, ctx->ev_cu, 0));
    }
    HIPCHK(hipMemsetAsync(ctx->counters.p, 0, 32 * sizeof(unsigned long long), s));
    HIPCHK(hipEventRecord(ctx->ev0, s));
    HIPCHK(geoac_launch_init(&P, s));
    ctx->n_epochs = 0; ctx->path_bytes_w = 0; ctx->path_bytes_r = 0;
    // late epochs (few waves still alive) are a quarter as long: the post-pass and the serial per-ray sums of the LAST epoch are the
    // uncovered tail of a fan, and both scale with the rows of that epoch
    const int rows_late = (ctx->s_rows_override >= 8) ? P.s_rows : std::max(std::min(1024, P.s_rows), P.s_rows / 4);
    int rows_now = P.s_rows;
    const long long max_epochs = (P.step_limit * (long long)ctx->legs) / (rows_late > 4 ? (rows_late - 3) : 1) + ctx->legs + 2;
    unsigned long long live = 1, live_bound = (unsigned long long)P.n_pad;
    // post-pass of one epoch on the second stream; gate_expected > 0: only after that many RK4 workgroups of this fan are resident
    auto enqueue_post = [&](GeoacDevParams Pq, size_t e, unsigned long long gate_expected) -> int {
        HIPCHK(hipStreamWaitEvent(sp, ctx->evs[4 * e + 1], 0));
        if(gate_expected > 0 && sp != s && !ctx->no_gate && !cu_split) HIPCHK(geoac_launch_gate(&Pq, gate_expected, sp));
        HIPCHK(hipEventRecord(ctx->evs[4 * e + 2], sp));
        if(Pq.atab_on){
            // (see geoac_launch_postpass_tab) hybrid fans: the post-pass off the RK4 CUs, one (spherical set) or two (Cartesian sets) workgroups per free CU
            Pq.pp_lds_pad = !Pq.table_in_lds ? 0 : (hybrid ? (Pq.eqset == GEOAC_EQ_GLOBAL ? 96 : 64) * 1024 : 8 * 1024);
            // fans of the spherical set that fill the chip: TWO post-pass workgroups per CU (two waves per SIMD).  Since round 4 the kernel needs 146 registers (three waves per
            // SIMD; 127 with the table entry in LDS: four) - and config 3 is SLOWER the more of them are resident: 358 ms per pass at two waves per SIMD, 378 at three, 385
            // at four, 369 at one (profiles/r04_c_cfg3_occupancy.txt).  RK4 (one wave per SIMD, 442 registers) and the post-pass cannot share a SIMD; every SIMD a
            // post-pass wave sits on is one an RK4 wave of the next round is not placed on, and RK4 is the longer of the two.
            if(!hybrid && Pq.eqset == GEOAC_EQ_GLOBAL) Pq.pp_lds_pad = 60 * 1024;
            if(cu_split) Pq.pp_lds_pad = 0;                   // (its own CUs: as many waves as its registers allow)
            if(ctx->pp_lds_pad >= 0) Pq.pp_lds_pad = ctx->pp_lds_pad;
            Pq.pp_onetrip = ctx->pp_onetrip >= 0 ? (ctx->pp_onetrip ? 1 : 0) : (hybrid ? 0 : 1);
            Pq.pp_lds_table = (Pq.eqset == GEOAC_EQ_GLOBAL && Pq.pp_onetrip) ? (ctx->pp_lds_table > 0 ? 1 : 0) : 0;      // (opt-in: measured no faster at any occupancy, see above)
            HIPCHK(hipMemsetAsync((char*)ctx->counters.p + GEOAC_CNT_PPFLAG * sizeof(unsigned long long), 0, sizeof(unsigned long long), sp));   // the fix-up list is empty
            HIPCHK(geoac_launch_postpass_tab(&Pq, Pq.s_rows, sp));
        }
        else HIPCHK(geoac_launch_postpass(&Pq, Pq.s_rows, sp));
        // the per-ray running sums (one thread per ray, latency-bound, 106 VGPRs: fits beside an RK4 wave) on their own stream, in
        // epoch order, so that the next epoch's post-pass does not queue behind them
        if(sa != sp){
            while(ctx->evp.size() < e + 1){ hipEvent_t ev; HIPCHK(hipEventCreate(&ev)); ctx->evp.push_back(ev); }
            HIPCHK(hipEventRecord(ctx->evp[e], sp));
            HIPCHK(hipStreamWaitEvent(sa, ctx->evp[e], 0));
        }
        HIPCHK(geoac_launch_accum(&Pq, sa));
        HIPCHK(hipEventRecord(ctx->evs[4 * e + 3], sa));
        return GEOAC_OK;
    };
    // RK4 workgroups that can be resident at once (k_rk4 runs one wave per SIMD; with the table in LDS one workgroup per CU)
    const unsigned wg_room = P.table_in_lds ? 192u : (unsigned)(768 / (block / 64));
    unsigned waves_launched_prev = 0, waves_launched_cur = 0;      // (GEOAC_TRACE_EPOCHS)
    unsigned long long wg_seen = 0;                   // RK4 workgroups launched in the earlier epochs of this fan
    GeoacDevParams Pprev = P;
    // The host never holds the GPU up between epochs: RK4(e) is enqueued BEFORE the live-ray count of epoch e-1 is read.  When that
    // count turns out to be zero the extra launch has found only finished rays (its workgroups return before staging the table).
    for(size_t e = 0; ; e++){
        const size_t eb = 4 * e;
        const int b = (int)(e % (size_t)n_chunks);
        while(ctx->evs.size() < eb + 4){ hipEvent_t ev; HIPCHK(hipEventCreate(&ev)); ctx->evs.push_back(ev); }
        while(ctx->evj.size() < 2 * e + 2){ hipEvent_t ev; HIPCHK(hipEventCreate(&ev)); ctx->evj.push_back(ev); }
        GeoacDevParams Pe = P;
        Pe.s_rows = rows_now;
        Pe.accum_batch = (ctx->accum_batch >= 0) ? (ctx->accum_batch ? 1 : 0) : (((rows_now == rows_late && rows_late != P.s_rows) || stagger_plan) ? 1 : 0);   // (few waves alive: the sums are the tail; staggered fans: measured 0-2 % faster in every epoch)
        if(late_pair) Pe.lanes_per_ray = 2;
        Pe.path = (double*)ctx->path[b].p; Pe.contrib = (double*)ctx->contrib[b].p;
        Pe.nrows = (int*)ctx->nrows[b].p; Pe.legend = (int*)ctx->legend[b].p; Pe.nlegend = (int*)ctx->nlegend[b].p;
        if(sampling){ Pe.ev_row = (int*)ctx->ev_row[b].p; Pe.ev_m = (int*)ctx->ev_m[b].p; Pe.ev_amp = (double*)ctx->ev_amp[b].p; Pe.nev = (int*)ctx->nev[b].p; }
        if(e >= (size_t)n_chunks) HIPCHK(hipStreamWaitEvent(s, ctx->evs[4 * (e - n_chunks) + 3], 0));      // chunk b free again?
        if(compact && e >= 1){
            // columns of this epoch = rays alive after the previous one; the launch is sized by the newest live count the host has
            // (after epoch e-2: an upper bound), lanes beyond the device-side count leave at once
            const int pb = (int)((e - 1) % (size_t)n_chunks);
            HIPCHK(geoac_launch_compact(&P, e == 1 ? nullptr : (const int*)ctx->colmap[pb].p, (const int*)ctx->ncols.p + pb, P.n_pad,
                                        (int*)ctx->colmap[b].p, (int*)ctx->ncols.p + b, s));
            Pe.colmap = (const int*)ctx->colmap[b].p; Pe.n_cols = (const int*)ctx->ncols.p + b;
            Pe.n_cols_bound = (int)std::min<unsigned long long>((unsigned long long)P.n_pad, (live_bound + 63ull) / 64ull * 64ull);
            if(Pe.n_cols_bound < 64) Pe.n_cols_bound = 64;
            Pe.slot_lo = 0; Pe.slot_hi = Pe.n_cols_bound;
        }
        HIPCHK(hipMemsetAsync((char*)ctx->counters.p + sizeof(unsigned long long), 0, sizeof(unsigned long long), s));
        HIPCHK(hipMemsetAsync((char*)ctx->counters.p + 4 * sizeof(unsigned long long), 0, sizeof(unsigned long long), s));
        HIPCHK(hipMemsetAsync((char*)ctx->counters.p + 6 * sizeof(unsigned long long), 0, 2 * sizeof(unsigned long long), s));
        HIPCHK(hipEventRecord(ctx->evs[eb], s));
        unsigned n_wg = 0, n_wg1 = 0;
        if(split){
            GeoacDevParams P1 = Pe;                   // one lane per ray, fewer rows: on the second RK4 stream
            P1.lanes_per_ray = 1; P1.slot_lo = n_pair; P1.slot_hi = P.n_pad; P1.live_slot = 6;
            P1.s_rows = std::max(8, (int)(plan_hybrid_rows * rows_now));
            HIPCHK(hipStreamWaitEvent(ctx->rk4b_stream, ctx->evs[eb], 0));
            HIPCHK(geoac_launch_rk4(&P1, 256, ctx->rk4b_stream, &n_wg1));
            HIPCHK(hipEventRecord(ctx->evj[2 * e], ctx->rk4b_stream));
            Pe.slot_lo = 0; Pe.slot_hi = n_pair;
            HIPCHK(geoac_launch_rk4(&Pe, 256, s, &n_wg));
            HIPCHK(hipStreamWaitEvent(s, ctx->evj[2 * e], 0));
            Pe.slot_hi = P.n_pad;
        } else if(stagger && (int)((long_bound + 255ull) / 256ull * 256ull) < Pe.slot_hi && long_bound > 0){
            const int kb = (int)((long_bound + 255ull) / 256ull * 256ull);       // columns [0, kb): the epoch's rows; [kb, slot_hi): fewer, on the second stream
            GeoacDevParams P1 = Pe;
            P1.slot_lo = kb; P1.live_slot = 6;
            P1.s_rows = std::max(8, (int)(plan_stagger_rows * rows_now));
            HIPCHK(hipStreamWaitEvent(ctx->rk4b_stream, ctx->evs[eb], 0));
            HIPCHK(geoac_launch_rk4(&P1, block, ctx->rk4b_stream, &n_wg1));
            HIPCHK(hipEventRecord(ctx->evj[2 * e], ctx->rk4b_stream));
            const int hi = Pe.slot_hi;
            Pe.slot_hi = kb;
            HIPCHK(geoac_launch_rk4(&Pe, block, s, &n_wg));
            HIPCHK(hipStreamWaitEvent(s, ctx->evj[2 * e], 0));
            Pe.slot_hi = hi;
        } else {
            // cooperative grid kernels with more waves than the chip has wave slots (1024: one per SIMD): sub-epochs (k_rk4) keep the
            // last round of the launch from running on a part-empty chip
            const int waves = (Pe.slot_hi - Pe.slot_lo + 63) / 64;
            if(P.coop && block == 64 && ctx->sub_epochs > 1 && waves > ctx->sub_min_waves && Pe.s_rows >= 16 * ctx->sub_epochs){
                Pe.sub = ctx->sub_epochs; Pe.sub_w = (waves + 7) / 8 * 8; Pe.sub_test_stall = ctx->sub_test_stall ? 1 : 0;
                HIPCHK(ctx->sub_flags.ensure(sizeof(int) * ((size_t)P.n_pad / 64 + 16)));
                Pe.sub_flags = (int*)ctx->sub_flags.p;
            }
            HIPCHK(geoac_launch_rk4(&Pe, block, s, &n_wg));
            Pe.sub = 1;                                // (Pe goes on to the post-pass / sum launches)
        }
        HIPCHK(hipEventRecord(ctx->evs[eb + 1], s));
        HIPCHK(hipMemcpyAsync(ctx->h_counters + 16 * (e & 1), ctx->counters.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(ctx->evj[2 * e + 1], s));
        // the post-pass of the previous epoch goes behind this epoch's RK4 workgroups (see k_gate)
        if(e >= 1){ int rc = enqueue_post(Pprev, e - 1, wg_seen + std::min(n_wg + n_wg1, wg_room)); if(rc != GEOAC_OK) return rc; }
        wg_seen += n_wg + n_wg1;
        waves_launched_prev = waves_launched_cur; waves_launched_cur = (n_wg + n_wg1) * (unsigned)((split ? 256 : block) / 64);
        Pprev = Pe;
        if(e >= 1){
            HIPCHK(hipEventSynchronize(ctx->evj[2 * (e - 1) + 1]));
            const unsigned long long* hc = ctx->h_counters + 16 * ((e - 1) & 1);
            live = hc[1] + hc[6];
            live_bound = live;
            if(hc[2] & 8ull){ hipDeviceSynchronize(); return fail(ctx, GEOAC_E_HIP, "k_rk4_duo: a wave waited more than a second for the other wave of its pair"); }
            if(live == 0){ ctx->n_epochs = e; break; }
            // hybrid fan: once the rays still alive would fit on half of the SIMDs as two-lane waves (hc[4] two-lane waves alive, hc[7]
            // one-lane waves that would become two each), or the two-lane share has finished, everything continues on the two-lane
            // kernel, whole epochs (same state layout): whatever is still running now sets the finish time - also when the shallow
            // rays were NOT the longest ones
            if(split && (hc[1] == 0 || hc[4] + 2 * hc[7] <= 512)) split = false;
            if(stagger){ long_bound = hc[1]; if(hc[4] + hc[7] <= 1024) stagger = false; }
            if(late_pair_able && !stagger && !late_pair && 2 * (hc[4] + hc[7]) <= 768) late_pair = true;     // (live one-lane waves; as two-lane waves they take three quarters of the chip at most: the post-pass of the epochs before keeps the rest)      // (everything fits one round now: one launch, whole epochs)
            if(hc[4] + hc[7] <= 256) rows_now = rows_late;
            if(ctx->trace_epochs) fprintf(stderr, "[epoch %zu] waves launched %u, after it: live rays %llu + %llu in %llu + %llu waves, split %d, rows next %d\n", e - 1, waves_launched_prev, hc[1], hc[6], hc[4], hc[7], (int)split, rows_now);
        }
        if((long long)e > max_epochs) return fail(ctx, GEOAC_E_CAPACITY, "fan_launch: epoch bound exceeded");
    }
    HIPCHK(geoac_launch_arrival(&P, s));           // inclination, back azimuth, range, amplitude of every arrival (k_arrival), beside the last post-pass
    HIPCHK(hipStreamWaitEvent(s, ctx->evs[4 * (ctx->n_epochs - 1) + 3], 0));
    HIPCHK(hipEventRecord(ctx->ev1, s));
    HIPCHK(hipMemcpyAsync(ctx->h_counters + 8, ctx->counters.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    ctx->n_samples = ctx->h_counters[8 + 3];
    {   unsigned long long fx = 0;
        HIPCHK(hipMemcpy(&fx, (const unsigned long long*)ctx->counters.p + GEOAC_CNT_PPFLAG + 1, sizeof(fx), hipMemcpyDeviceToHost));
        ctx->pp_fixup_segments = fx; }
    if(ctx->trace_epochs && P.trio){
        unsigned long long v = 0;
        HIPCHK(hipMemcpy(&v, (const unsigned long long*)ctx->counters.p + 30, sizeof(v), hipMemcpyDeviceToHost));
        fprintf(stderr, "[trio] workgroups whose three waves sat on three SIMDs: %llu, others: %llu\n", v & 0xffffffffull, v >> 32);
    }
#ifdef GEOAC_KSTAT
    {   // diagnostic build (geoac_rngdep.h): distinct (segment, cell) keys per live wave-stage
        unsigned long long h[16];
        HIPCHK(hipMemcpy(h, (const unsigned long long*)ctx->counters.p + 16, sizeof(h), hipMemcpyDeviceToHost));
        unsigned long long tot = 0; for(int i = 0; i < 8; i++) tot += h[i];
        if(tot){
            fprintf(stderr, "[kstat] wave-stages %llu; K = 1: %.3f, 2: %.3f, 3-4: %.3f, 5-6: %.3f, 7-8: %.3f, 9-12: %.3f, 13-16: %.3f, > 16: %.3f; mean K %.2f, mean kz span %.2f\n", tot,
                    (double)h[0] / tot, (double)h[1] / tot, (double)h[2] / tot, (double)h[3] / tot, (double)h[4] / tot, (double)h[5] / tot, (double)h[6] / tot, (double)h[7] / tot,
                    (double)h[8] / tot, (double)h[9] / tot);
            if(h[11]) fprintf(stderr, "[kstat] lane-stages %llu, of them with a (segment, cell) key other than the lane's key of the stage before: %.4f\n", h[11], (double)h[10] / (double)h[11]);
        }
    }
#endif

    float ms = 0; hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->ms_total = ms;
    ctx->ms_rk4 = 0; ctx->ms_post = 0;
    for(size_t e = 0; e < (size_t)ctx->n_epochs; e++){
        float a = 0, b = 0;
        hipEventElapsedTime(&a, ctx->evs[4 * e], ctx->evs[4 * e + 1]);
        hipEventElapsedTime(&b, ctx->evs[4 * e + 2], ctx->evs[4 * e + 3]);
        ctx->ms_rk4 += a; ctx->ms_post += b;
    }
    ctx->total_steps = ctx->h_counters[8 + 0];
    ctx->err_flags = ctx->h_counters[8 + 2];
    // algorithmic path traffic: one PATHW-wide row per step (+ leg-start / carry rows, not counted)
    ctx->path_bytes_w = ctx->total_steps * (unsigned long long)(P.pathw * sizeof(double));
    ctx->path_bytes_r = 2 * ctx->path_bytes_w;
    ctx->ran = true;
    ctx->lastP = P;
    if(ctx->err_flags & 4ull){
        // a sub-epoch workgroup gave up waiting for its predecessor: the dispatch order k_rk4 relies on did not hold.  The fan is run again
        // without sub-epochs; this context keeps them off from now on and says so in geoac_fan_status (GEOAC_FAN_SUB_FALLBACK)
        if(ctx->sub_epochs > 1){ ctx->sub_epochs = 1; ctx->sticky_flags |= GEOAC_FAN_SUB_FALLBACK; return GEOAC_RETRY; }
        return fail(ctx, GEOAC_E_HIP, "sub-epoch hand-off timed out");
    }
    if(ctx->err_flags & 16ull){
        // more segments outside the absorption table than its fix-up list holds: this context keeps the exact post-pass from now on (GEOAC_FAN_ABS_FALLBACK)
        if(ctx->abs_table){ ctx->abs_table = 0; ctx->sticky_flags |= GEOAC_FAN_ABS_FALLBACK; return GEOAC_RETRY; }
        return fail(ctx, GEOAC_E_CAPACITY, "fix-up list of the table post-pass overflowed");
    }
    if(ctx->err_flags & 8ull) return fail(ctx, GEOAC_E_HIP, "k_rk4_duo: a wave waited more than a second for the other wave of its pair");
    if(ctx->err_flags & 2ull) return fail(ctx, GEOAC_E_CAPACITY, "per-epoch sample/caustic event list overflowed");
    if(sampling && ctx->n_samples > (unsigned long long)P.smp_cap)
        return fail(ctx, GEOAC_E_CAPACITY, "sample list overflowed: raise GEOAC_SMP_CAP (needed " + std::to_string(ctx->n_samples) + ")");
    // a ray that exhausts step_limit is an ordinary leg end for the reference (GeoAc_Propagate_RK4 returns step_limit with check = false
    // and the mains write the row): the fan is complete and valid, the condition is reported through geoac_fan_status
    if(ctx->err_flags & 1ull) ctx->err = "warning: a ray reached step_limit (GeoAc.Solver.cpp:14) without leaving the region or reaching the ground";
    return GEOAC_OK;
}

int geoac_fan_launch(geoac_ctx* ctx){
    if(!ctx) return GEOAC_E_INVALID;
    if(ctx->src_gen && ctx->src_gen->load() != ctx->src_gen_at_clone)
        return fail(ctx, GEOAC_E_INVALID, "fan_launch: this context is a clone (geoac_clone) and its source has uploaded another atmosphere or was destroyed since - "
                                          "the clone's views of the source's tables are no longer valid; clone again");
    struct Guard { Guard(){ g_launching.fetch_add(1); } ~Guard(){ if(g_launching.fetch_sub(1) == 1) g_deferred.drain_if_any(); } } guard;
    // at most one repeat per plan feature that can be withdrawn (sub-epochs, absorption table): a loop, not a recursion
    for(int attempt = 0; attempt < 3; attempt++){
        const int rc = fan_launch_once(ctx);
        if(rc != GEOAC_RETRY) return rc;
        ctx->launch_repeats++;
    }
    return fail(ctx, GEOAC_E_HIP, "fan_launch: still asked to repeat after withdrawing every optional plan feature");
}

int geoac_abs_table_info(geoac_ctx* ctx, int* entries, int* flagged, uint64_t* fixup_segments, double* worst_rel_err){
    if(!ctx || !ctx->ran) return GEOAC_E_INVALID;
    if(worst_rel_err) *worst_rel_err = ctx->lastP.atab_on ? ctx->atab_worst : 0.0;
    if(entries) *entries = ctx->lastP.atab_on ? ctx->atab_entries : 0;
    if(flagged) *flagged = ctx->lastP.atab_on ? ctx->atab_flagged : 0;
    if(fixup_segments) *fixup_segments = (uint64_t)ctx->pp_fixup_segments;
    return GEOAC_OK;
}

int geoac_fan_status(geoac_ctx* ctx, uint64_t* flags){
    if(!ctx || !flags || !ctx->ran) return GEOAC_E_INVALID;
    // bit 0 of the launch's own flags (the other device-side bits are failures, reported as errors) + what this context has withdrawn from its plan
    *flags = (uint64_t)(ctx->err_flags & 1ull) | (uint64_t)ctx->sticky_flags;
    return GEOAC_OK;
}

int geoac_fan_sync(geoac_ctx* ctx){
    if(!ctx) return GEOAC_E_INVALID;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GEOAC_OK;
}

int geoac_fan_records_dev(geoac_ctx* ctx, void** dev_ptr, size_t* bytes){
    if(!ctx || !ctx->ran) return fail(ctx, GEOAC_E_INVALID, "fan_records_dev: no completed launch");
    if(dev_ptr) *dev_ptr = ctx->rec.p;
    if(bytes) *bytes = sizeof(double) * (size_t)ctx->n_rays * ctx->legs * GEOAC_REC_STRIDE;
    return GEOAC_OK;
}

int geoac_fan_copy_records_dev(geoac_ctx* ctx, void* dst_dev){
    if(!ctx || !ctx->ran || !dst_dev) return fail(ctx, GEOAC_E_INVALID, "fan_copy_records_dev: no completed launch / null destination");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(dst_dev, ctx->rec.p, sizeof(double) * (size_t)ctx->n_rays * ctx->legs * GEOAC_REC_STRIDE,
                          hipMemcpyDeviceToDevice, ctx->stream));
    return GEOAC_OK;
}

int geoac_fan_fetch(geoac_ctx* ctx, double* rec_host, uint64_t* total_steps){
    if(!ctx || !ctx->ran) return fail(ctx, GEOAC_E_INVALID, "fan_fetch: no completed launch");
    HIPCHK(hipSetDevice(ctx->device));
    if(rec_host){
        HIPCHK(hipMemcpyAsync(rec_host, ctx->rec.p, sizeof(double) * (size_t)ctx->n_rays * ctx->legs * GEOAC_REC_STRIDE,
                              hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    if(total_steps) *total_steps = ctx->total_steps;
    return GEOAC_OK;
}

int geoac_fan_sample_count(geoac_ctx* ctx, int64_t* n){
    if(!ctx || !n || !ctx->ran) return GEOAC_E_INVALID;
    *n = (int64_t)ctx->n_samples;
    return GEOAC_OK;
}

int geoac_fan_set_sample_capacity(geoac_ctx* ctx, int64_t rows){
    if(!ctx || rows < 1) return GEOAC_E_INVALID;
    ctx->smp_cap = rows;
    return GEOAC_OK;
}

int geoac_fan_fetch_samples(geoac_ctx* ctx, double* smp_host, int64_t cap){
    if(!ctx || !ctx->ran || !smp_host) return fail(ctx, GEOAC_E_INVALID, "fan_fetch_samples: no completed launch / null buffer");
    int64_t n = (int64_t)ctx->n_samples;
    if(n > ctx->smp_cap) n = ctx->smp_cap;
    if(cap < n) return fail(ctx, GEOAC_E_CAPACITY, "fan_fetch_samples: buffer too small");
    if(n == 0) return GEOAC_OK;
    HIPCHK(hipSetDevice(ctx->device));
    std::unique_ptr<double[]> tmp(new double[(size_t)n * GEOAC_SMP_STRIDE]);              // (not zeroed: 80 B x up to 16 Mi rows)
    HIPCHK(hipMemcpyAsync(tmp.get(), ctx->smp_out.p, sizeof(double) * (size_t)n * GEOAC_SMP_STRIDE, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // The device list is in completion order; the files are in (ray, leg, m) order, raypath rows before caustic rows.  One thread of k_accum emits a ray's rows of an
    // epoch in that order and the epochs' launches are stream-ordered, so the rows of ONE ray already arrive sorted: a stable counting sort by ray (O(n)) puts the table
    // in order; every ray's run is then checked and, should it ever not be sorted, sorted on its own.  (Until round 4: one std::sort of all rows through an indirect
    // four-key comparison - 2 s per 8 M rows, the largest single cost of a WriteRays run once the text was formatted in parallel.)
    const double* t = tmp.get();
    const int64_t nray = (int64_t)ctx->n_rays;
    std::vector<int64_t> start((size_t)nray + 1, 0);
    for(int64_t i = 0; i < n; i++){
        const int64_t r = (int64_t)t[i * GEOAC_SMP_STRIDE + GEOAC_SMP_RAY];
        if(r < 0 || r >= nray) return fail(ctx, GEOAC_E_HIP, "fan_fetch_samples: a sample row names a ray outside the fan");
        start[(size_t)r + 1]++;
    }
    for(int64_t r = 0; r < nray; r++) start[(size_t)r + 1] += start[(size_t)r];
    std::vector<int64_t> idx((size_t)n), fill(start.begin(), start.end() - 1);
    for(int64_t i = 0; i < n; i++) idx[(size_t)fill[(size_t)(int64_t)t[i * GEOAC_SMP_STRIDE + GEOAC_SMP_RAY]]++] = i;
    auto less = [t](int64_t a, int64_t b){
        const double* A = t + a * GEOAC_SMP_STRIDE; const double* B = t + b * GEOAC_SMP_STRIDE;
        for(int q = 0; q < 4; q++){ if(A[q] != B[q]) return A[q] < B[q]; }
        return a < b;
    };
    // check / repair and copy out, rays dealt to a few threads (the copy is 80-byte rows from all over a table of up to 1.3 GB)
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(8, n / 200000));
    auto work = [&](int w){
        const int64_t r0 = nray * w / nt, r1 = nray * (w + 1) / nt;
        for(int64_t r = r0; r < r1; r++){
            int64_t* b = idx.data() + start[(size_t)r]; int64_t* e = idx.data() + start[(size_t)r + 1];
            if(!std::is_sorted(b, e, less)) std::sort(b, e, less);
        }
        for(int64_t i = start[(size_t)r0]; i < start[(size_t)r1]; i++)
            memcpy(smp_host + i * GEOAC_SMP_STRIDE, t + idx[(size_t)i] * GEOAC_SMP_STRIDE, sizeof(double) * GEOAC_SMP_STRIDE);
    };
    std::vector<std::thread> pool;
    for(int w = 1; w < nt; w++) pool.emplace_back(work, w);
    work(0);
    for(auto& th : pool) th.join();
    return GEOAC_OK;
}

int geoac_fan_run(geoac_ctx* ctx, int n_rays, const double* theta_deg, const double* phi_deg,
                  double* rec_host, uint64_t* total_steps){
    int rc = geoac_fan_set_angles(ctx, n_rays, theta_deg, phi_deg);
    if(rc) return rc;
    rc = geoac_fan_launch(ctx);
    if(rc) return rc;
    return geoac_fan_fetch(ctx, rec_host, total_steps);
}

// ---- device-function probes (include/geoac_probe.h) ----
namespace {
struct ProbeBufs {
    std::vector<void*> d;
    ~ProbeBufs(){ for(void* p : d) if(p) hipFree(p); }
    double* in(const double* h, size_t n){ void* p = nullptr; if(hipMalloc(&p, n * sizeof(double)) != hipSuccess) return nullptr; d.push_back(p);
                                           if(hipMemcpy(p, h, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return nullptr; return (double*)p; }
    double* out(size_t n){ void* p = nullptr; if(hipMalloc(&p, n * sizeof(double)) != hipSuccess) return nullptr; d.push_back(p); return (double*)p; }
};
}

int geoac_probe_atmo_1d(geoac_ctx* ctx, int n, const double* x, double* out9, double* rho){
    if(!ctx || n <= 0 || !x || !out9 || !rho) return fail(ctx, GEOAC_E_INVALID, "probe_atmo_1d: bad arguments");
    if(!ctx->ran || ctx->have_grid) return fail(ctx, GEOAC_E_INVALID, "probe_atmo_1d: needs a 1-D equation set and a completed launch");
    HIPCHK(hipSetDevice(ctx->device));
    ProbeBufs B; double* dx = B.in(x, (size_t)n); double* d9 = B.out((size_t)9 * n); double* dr = B.out((size_t)n);
    if(!dx || !d9 || !dr) return fail(ctx, GEOAC_E_NOMEM, "probe: device allocation failed");
    HIPCHK(geoac_launch_probe_atmo1d(&ctx->lastP, n, dx, d9, dr, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(out9, d9, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rho, dr, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    return GEOAC_OK;
}

int geoac_probe_absorption(geoac_ctx* ctx, int n, const double* x, const double* freq, double* alpha){
    if(!ctx || n <= 0 || !x || !freq || !alpha) return fail(ctx, GEOAC_E_INVALID, "probe_absorption: bad arguments");
    if(!ctx->ran || ctx->have_grid) return fail(ctx, GEOAC_E_INVALID, "probe_absorption: needs a 1-D equation set and a completed launch");
    HIPCHK(hipSetDevice(ctx->device));
    ProbeBufs B; double* dx = B.in(x, (size_t)n); double* df = B.in(freq, (size_t)n); double* da = B.out((size_t)n);
    if(!dx || !df || !da) return fail(ctx, GEOAC_E_NOMEM, "probe: device allocation failed");
    HIPCHK(geoac_launch_probe_absorption(&ctx->lastP, n, dx, df, da, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(alpha, da, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    return GEOAC_OK;
}

int geoac_probe_absorption_table(geoac_ctx* ctx, int n, const double* x, double* alpha){
    if(!ctx || n <= 0 || !x || !alpha) return fail(ctx, GEOAC_E_INVALID, "probe_absorption_table: bad arguments");
    if(!ctx->ran || ctx->have_grid || !ctx->lastP.atab_on) return fail(ctx, GEOAC_E_INVALID, "probe_absorption_table: needs a completed launch of a 1-D set with the absorption table on");
    HIPCHK(hipSetDevice(ctx->device));
    ProbeBufs B; double* dx = B.in(x, (size_t)n); double* da = B.out((size_t)n);
    if(!dx || !da) return fail(ctx, GEOAC_E_NOMEM, "probe: device allocation failed");
    HIPCHK(geoac_launch_probe_atab(&ctx->lastP, n, dx, da, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(alpha, da, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    return GEOAC_OK;
}

int geoac_probe_grid(geoac_ctx* ctx, int n, const double* a0, const double* a1, const double* a2, int coop, double* out30, double* api7){
    if(!ctx || n <= 0 || !a0 || !a1 || !a2 || !out30 || !api7) return fail(ctx, GEOAC_E_INVALID, "probe_grid: bad arguments");
    if(!ctx->ran || !ctx->have_grid) return fail(ctx, GEOAC_E_INVALID, "probe_grid: needs a grid equation set and a completed launch");
    HIPCHK(hipSetDevice(ctx->device));
    ProbeBufs B; double* d0 = B.in(a0, (size_t)n); double* d1 = B.in(a1, (size_t)n); double* d2 = B.in(a2, (size_t)n);
    double* do30 = B.out((size_t)30 * n); double* da7 = B.out((size_t)7 * n);
    if(!d0 || !d1 || !d2 || !do30 || !da7) return fail(ctx, GEOAC_E_NOMEM, "probe: device allocation failed");
    HIPCHK(geoac_launch_probe_grid(&ctx->lastP, n, d0, d1, d2, coop ? 1 : 0, do30, da7, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(out30, do30, sizeof(double) * 30 * (size_t)n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(api7, da7, sizeof(double) * 7 * (size_t)n, hipMemcpyDeviceToHost));
    return GEOAC_OK;
}

int geoac_last_timing(geoac_ctx* ctx, double ms[3], uint64_t stats[3]){
    if(!ctx || !ctx->ran) return GEOAC_E_INVALID;
    if(ms){ ms[0] = ctx->ms_total; ms[1] = ctx->ms_rk4; ms[2] = ctx->ms_post; }
    if(stats){ stats[0] = ctx->n_epochs; stats[1] = ctx->path_bytes_w; stats[2] = ctx->path_bytes_r; }
    return GEOAC_OK;
}

}  // extern "C"
