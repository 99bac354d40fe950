// CPU-only ThreadSanitizer driver (tests/test_sanitizers.py): the eigenray scheduler of libgeoac_hip (geoac_amd/csrc/geoac_eigenray.cpp:
// tasks advanced in rounds by the caller's thread; the (bounces, CalcAmp, mode) groups of a round integrated side by side, one host thread
// and one context clone each) compiled with -fsanitize=thread on top of a STUB of the fan ABI - a closed-form "ray" (arrival range grows
// with the inclination, bearing = launch azimuth) in place of the GPU.  No physics is checked here: the point is the round / worker-thread /
// deferred-log machinery under TSan.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/geoac_eig.h"

#include <atomic>
struct geoac_ctx { geoac_params p; int eqset; std::string err; long calls; };
static std::atomic<long> g_clone_calls{0};
static const double Pi = 3.141592653589793;

extern "C" {
int geoac_default_params(int eqset, geoac_params* p){ memset(p, 0, sizeof *p); p->ds_min = 0.001; p->ds_max = 0.5; p->ray_limit = 5000; p->tweak_abs = 0.3; p->freq = 0.1;
    p->bounces = 2; p->calc_amp = 1; p->sample_stride = 25; p->vert_limit = 139.9; p->range_limit = 10000; (void)eqset; return 0; }
int geoac_get_params(geoac_ctx* c, geoac_params* p){ *p = c->p; return 0; }
int geoac_set_params(geoac_ctx* c, const geoac_params* p){ c->p = *p; return 0; }
int geoac_get_eqset(geoac_ctx* c, int* e){ *e = c->eqset; return 0; }
// clones: the groups of a decision round are integrated side by side, one context and one host thread each (serve)
int geoac_clone(geoac_ctx* src, geoac_ctx** out){ geoac_ctx* c = new geoac_ctx(); c->p = src->p; c->eqset = src->eqset; c->calls = 0; *out = c; return 0; }
int geoac_destroy(geoac_ctx* c){ g_clone_calls += c->calls; delete c; return 0; }
int geoac_medium_1d(geoac_ctx*, double, double out[4]){ out[0] = 0.34; out[1] = 0.0; out[2] = 0.0; out[3] = 1.2e-3; return 0; }
int geoac_fan_sample_count(geoac_ctx*, int64_t* n){ *n = 0; return 0; }
int geoac_fan_fetch_samples(geoac_ctx*, double*, int64_t){ return 0; }
const char* geoac_last_error(geoac_ctx* c){ return c->err.c_str(); }
const char* geoac_strerror(int){ return "stub"; }
// the stand-in for the GPU: leg l of a ray launched at (theta, azimuth-from-north phi) comes down (l + 1) (150 + 8 theta) km away
int geoac_fan_run(geoac_ctx* c, int n, const double* th, const double* ph, double* rec, uint64_t* steps){
    const int legs = c->p.bounces + 1;
    memset(rec, 0, sizeof(double) * (size_t)n * legs * GEOAC_REC_STRIDE);
    for(int i = 0; i < n; i++){
        const double a = (90.0 - ph[i]) * Pi / 180.0, R1 = 150.0 + 8.0 * th[i], dR = 8.0 * 180.0 / Pi;
        for(int l = 0; l < legs; l++){
            double* R = rec + ((size_t)i * legs + l) * GEOAC_REC_STRIDE; double* S = R + GEOAC_REC_STATE;
            const double r = (l + 1) * R1;
            R[GEOAC_REC_VALID] = 1; R[GEOAC_REC_STEPS] = 1000 + l; R[GEOAC_REC_TTIME] = r / 0.3; R[GEOAC_REC_ATTEN] = 0.01 * r; R[GEOAC_REC_TURN] = 40;
            R[GEOAC_REC_INCL] = th[i]; R[GEOAC_REC_BACKAZ] = ph[i] + 180.0; R[GEOAC_REC_AMP] = 1.0 / (4 * Pi * r); R[GEOAC_REC_RANGE] = r; R[GEOAC_REC_JACOB] = r;
            S[0] = c->p.src[0] + r * cos(a); S[1] = c->p.src[1] + r * sin(a); S[2] = -1e-4; S[3] = -0.9;
            S[4] = (l + 1) * dR * cos(a); S[5] = (l + 1) * dR * sin(a); S[6] = 0; S[7] = 0;
            S[8] = -r * sin(a); S[9] = r * cos(a); S[10] = 0; S[11] = 0;
        }
    }
    if(steps) *steps = (uint64_t)n * 1000;
    c->calls++;
    return 0;
}
}

int main(){
    geoac_ctx ctx; ctx.eqset = GEOAC_EQ_3D; ctx.calls = 0;
    geoac_default_params(GEOAC_EQ_3D, &ctx.p);
    geoac_eig_params E; geoac_eig_default_params(&E);
    E.bnc_min = 0; E.bnc_max = 1; E.verbose = 1;
    std::vector<double> rcv;
    for(int k = 0; k < 12; k++){ const double az = k * 30.0 * Pi / 180.0, r = 220.0 + 23.0 * k; rcv.push_back(r * cos(az)); rcv.push_back(r * sin(az)); }
    geoac_eig_result* res = nullptr;
    int rc = geoac_eig_search(&ctx, &E, 12, rcv.data(), &res);
    if(rc || !res){ fprintf(stderr, "eig_search failed: %d\n", rc); return 1; }
    const long ne = (long)geoac_eig_count(res);
    uint64_t st[4]; geoac_eig_stats(res, st);
    size_t loglen = 0; for(int k = 0; k < 12; k++) loglen += strlen(geoac_eig_log(res, k));
    printf("tsan_eig_driver ok: %ld eigenrays, %llu launches, %llu rays, %llu rounds, %zu log bytes\n", ne, (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[3], loglen);
    geoac_eig_free(res);
    return ne >= 12 ? 0 : 1;
}
