// geoac_multi.cpp - a fan over several GPUs from one process: one geoac_ctx and one host thread per device, azimuth groups from a
// shared queue, records copied straight into the caller's table (include/geoac_multi.h).  Plain C++ on top of the C ABI of geoac_hip.h.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/geoac_multi.h"

struct geoac_pool {
    int eqset = 0;
    std::vector<geoac_ctx*> ctx;
    std::vector<uint64_t> rays, steps, groups;
    std::string err;
    uint64_t status = 0;                       // geoac_fan_status flags of the last run, OR-ed over every group
};

static int pool_fail(geoac_pool* p, int rc, const std::string& m){ if(p) p->err = m; return rc; }

extern "C" {

int geoac_pool_create(geoac_pool** out, int eqset, int n_dev, const int* devices){
    if(!out || n_dev <= 0 || !devices) return GEOAC_E_INVALID;
    geoac_pool* p = new geoac_pool();
    p->eqset = eqset;
    for(int i = 0; i < n_dev; i++){
        geoac_ctx* c = nullptr;
        int rc = geoac_create(&c, eqset, devices[i]);
        if(rc){ for(geoac_ctx* q : p->ctx) geoac_destroy(q); delete p; return rc; }
        p->ctx.push_back(c);
    }
    p->rays.assign((size_t)n_dev, 0); p->steps.assign((size_t)n_dev, 0); p->groups.assign((size_t)n_dev, 0);
    *out = p;
    return GEOAC_OK;
}

int geoac_pool_destroy(geoac_pool* p){
    if(!p) return GEOAC_E_INVALID;
    for(geoac_ctx* c : p->ctx) geoac_destroy(c);
    delete p;
    return GEOAC_OK;
}

int geoac_pool_size(const geoac_pool* p){ return p ? (int)p->ctx.size() : 0; }
geoac_ctx* geoac_pool_ctx(geoac_pool* p, int i){ return (p && i >= 0 && i < (int)p->ctx.size()) ? p->ctx[(size_t)i] : nullptr; }

int geoac_pool_upload_atmo_1d(geoac_pool* p, int n, const double* x, const double* T, const double* u, const double* v, const double* rho, const double* slopes4){
    if(!p) return GEOAC_E_INVALID;
    for(geoac_ctx* c : p->ctx){ int rc = geoac_upload_atmo_1d(c, n, x, T, u, v, rho, slopes4); if(rc) return pool_fail(p, rc, geoac_last_error(c)); }
    return GEOAC_OK;
}

int geoac_pool_upload_atmo_3d(geoac_pool* p, int nx, int ny, int nz, const double* x, const double* y, const double* z,
                              const double* T, const double* u, const double* v, const double* rho){
    if(!p) return GEOAC_E_INVALID;
    // the table build runs on each device (geoac_gridbuild.hip); the uploads are independent, so one thread per context
    std::vector<int> rcs(p->ctx.size(), 0);
    std::vector<std::thread> th;
    for(size_t i = 0; i < p->ctx.size(); i++) th.emplace_back([&, i]{ rcs[i] = geoac_upload_atmo_3d(p->ctx[i], nx, ny, nz, x, y, z, T, u, v, rho); });
    for(auto& t : th) t.join();
    for(size_t i = 0; i < rcs.size(); i++) if(rcs[i]) return pool_fail(p, rcs[i], geoac_last_error(p->ctx[i]));
    return GEOAC_OK;
}

int geoac_pool_set_params(geoac_pool* p, const geoac_params* prm){
    if(!p || !prm) return GEOAC_E_INVALID;
    if(prm->mode & (GEOAC_MODE_WRITE_RAYS | GEOAC_MODE_WRITE_CAUSTICS)) return pool_fail(p, GEOAC_E_UNSUPPORTED, "pool: sample capture (WriteRays / WriteCaustics) runs on a single context");
    for(geoac_ctx* c : p->ctx){ int rc = geoac_set_params(c, prm); if(rc) return pool_fail(p, rc, geoac_last_error(c)); }
    return GEOAC_OK;
}

int geoac_pool_fan_run(geoac_pool* p, int n_rays, const double* theta, const double* phi, int rays_per_group, double* rec_host, uint64_t* total_steps){
    if(!p || n_rays <= 0 || !theta || !phi || !rec_host) return pool_fail(p, GEOAC_E_INVALID, "pool_fan_run: bad arguments");
    const size_t D = p->ctx.size();
    // the record stride comes from the contexts themselves (their defaults or whatever was set through the pool or through
    // geoac_pool_ctx): every context must write the same number of legs per ray in the same mode, or the groups would overlap
    geoac_params p0{};
    if(geoac_get_params(p->ctx[0], &p0)) return pool_fail(p, GEOAC_E_INVALID, "pool_fan_run: context without parameters");
    for(size_t d = 1; d < D; d++){
        geoac_params pd{};
        geoac_get_params(p->ctx[d], &pd);
        if(pd.bounces != p0.bounces || pd.calc_amp != p0.calc_amp || pd.mode != p0.mode)
            return pool_fail(p, GEOAC_E_INVALID, "pool_fan_run: the pool's contexts hold different parameters (bounces / calc_amp / mode)");
    }
    if(p0.mode & (GEOAC_MODE_WRITE_RAYS | GEOAC_MODE_WRITE_CAUSTICS)) return pool_fail(p, GEOAC_E_UNSUPPORTED, "pool: sample capture (WriteRays / WriteCaustics) runs on a single context");
    const int legs = p0.bounces + 1;
    // azimuth boundaries: the reference's outer loop variable (rays are phi-major)
    std::vector<int> az;
    for(int i = 0; i < n_rays; i++) if(i == 0 || phi[i] != phi[i - 1]) az.push_back(i);
    az.push_back(n_rays);
    const int n_az = (int)az.size() - 1;
    long want = rays_per_group;
    if(want <= 0){
        want = (n_rays + (long)(4 * D) - 1) / (long)(4 * D);            // about four groups per device ...
        want = std::max(want, std::min<long>(16384, (n_rays + (long)D - 1) / (long)D));   // ... but full-sized launches while the fan allows
    }
    std::vector<std::pair<int, int>> groups;                            // [first ray, one past last ray)
    for(int a = 0; a < n_az; ){
        int b = a + 1;
        while(b < n_az && az[(size_t)b + 1] - az[(size_t)a] <= want) b++;
        groups.emplace_back(az[(size_t)a], az[(size_t)b]);
        a = b;
    }
    std::atomic<size_t> next{0};
    std::atomic<int> first_rc{0};
    std::mutex m;
    std::fill(p->rays.begin(), p->rays.end(), 0); std::fill(p->steps.begin(), p->steps.end(), 0); std::fill(p->groups.begin(), p->groups.end(), 0);
    const size_t row = (size_t)legs * GEOAC_REC_STRIDE;
    std::atomic<uint64_t> status{0};
    std::vector<std::thread> th;
    for(size_t d = 0; d < D; d++){
        th.emplace_back([&, d]{
            geoac_ctx* c = p->ctx[d];
            for(;;){
                const size_t g = next.fetch_add(1);
                if(g >= groups.size() || first_rc.load()) break;
                const int i0 = groups[g].first, i1 = groups[g].second;
                uint64_t st = 0;
                // records of the group land in their rows of the caller's table: this copy IS the gather
                int rc = geoac_fan_run(c, i1 - i0, theta + i0, phi + i0, rec_host + (size_t)i0 * row, &st);
                if(rc){
                    int expected = 0;
                    if(first_rc.compare_exchange_strong(expected, rc)){ std::lock_guard<std::mutex> lk(m); p->err = geoac_last_error(c); }
                    break;
                }
                uint64_t fl = 0;
                if(geoac_fan_status(c, &fl) == GEOAC_OK) status.fetch_or(fl);
                p->rays[d] += (uint64_t)(i1 - i0); p->steps[d] += st; p->groups[d] += 1;
            }
        });
    }
    for(auto& t : th) t.join();
    p->status = status.load();
    if(first_rc.load()) return first_rc.load();
    if(total_steps){ uint64_t s = 0; for(uint64_t v : p->steps) s += v; *total_steps = s; }
    return GEOAC_OK;
}

int geoac_pool_last_shares(const geoac_pool* p, uint64_t* rays, uint64_t* steps, uint64_t* groups){
    if(!p) return GEOAC_E_INVALID;
    for(size_t i = 0; i < p->ctx.size(); i++){ if(rays) rays[i] = p->rays[i]; if(steps) steps[i] = p->steps[i]; if(groups) groups[i] = p->groups[i]; }
    return GEOAC_OK;
}

int geoac_pool_fan_status(const geoac_pool* p, uint64_t* flags){
    if(!p || !flags) return GEOAC_E_INVALID;
    *flags = p->status;
    return GEOAC_OK;
}

const char* geoac_pool_last_error(const geoac_pool* p){ return p ? p->err.c_str() : "null pool"; }

}  // extern "C"
