// geoac_cli.cpp - host drivers `GeoAc2D`, `GeoAc3D`, `GeoAcGlobal`, `GeoAc3D.RngDep`, `GeoAcGlobal.RngDep` (built with -DGEOAC_CLI_SET=0/1/2/3/4):
// the reference's `-prop` command line, .met input and `_results.dat` / `_raypaths.dat` / `_caustics-path#.dat` /
// `atmo.dat` outputs, with the launch-angle fan integrated by libgeoac_hip.so on the GPU.
//
// Surface kept (grammar, defaults, ordering side effects, number formatting):
//   GeoAcGlobal_RunProp  Code/GeoAcGlobal_main.cpp:116-332      GeoAc3D_RunProp  Code/GeoAc3D_main.cpp:106-313
//   GeoAc2D_RunProp      Code/GeoAc2D_main.cpp:66-237           GeoAc_WriteProfile  Code/GeoAc/GeoAc.Interface{,.Global}.cpp:78-98
//   GeoAc3D_RngDep_RunProp  Code/GeoAc3D.RngDep_main.cpp:119-334 (`-prop prefix loc_x loc_y [parameter=value ...]`)
//   GeoAcGlobal_RngDep_RunProp  Code/GeoAcGlobal.RngDep_main.cpp:122-343 (`-prop prefix loc_lat loc_lon [parameter=value ...]`)
//   GeoAcGlobal_RunEigSearch / RunEigDirect (+ the .RngDep twins)  Code/GeoAcGlobal_main.cpp:496-660, Code/GeoAcGlobal.RngDep_main.cpp:518-693
//     (spherical mains only; the searches themselves are geoac_eig_search / geoac_eig_direct, include/geoac_eig.h)
//   -interactive of the five mains: run_interactive below.
// Environment (not part of the reference's surface): GEOAC_DEVICES=0,1,... integrates an arrivals-only -prop fan (WriteRays=False, no
// caustics) on several GPUs of the node through geoac_pool_fan_run (include/geoac_multi.h); GEOAC_STATS=<file> writes a JSON
// summary of the run (rays, RK4 ray-steps, seconds on the GPU, per-device shares).
#include <strings.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
#include <thread>
#include <vector>
#include <atomic>
#include <charconv>
#include <condition_variable>
#include <mutex>
#include <sstream>

#include "../../../include/geoac_hip.h"
#include "../../../include/geoac_host.h"
#include "../../../include/geoac_eig.h"
#include "../../../include/geoac_multi.h"
#include <chrono>

#ifndef GEOAC_CLI_SET
#define GEOAC_CLI_SET 2
#endif

using namespace std;

static const double Pi = 3.141592653589793238462643;
static const double kGamR = 0.00040187, kR = 287.05, kGam = 1.4;
static const char* kName = (GEOAC_CLI_SET == 0) ? "GeoAc2D" : (GEOAC_CLI_SET == 1) ? "GeoAc3D" : (GEOAC_CLI_SET == 2) ? "GeoAcGlobal"
                         : (GEOAC_CLI_SET == 3) ? "GeoAc3D.RngDep" : "GeoAcGlobal.RngDep";
static const int kEq = (GEOAC_CLI_SET == 0) ? GEOAC_EQ_2D : (GEOAC_CLI_SET == 1) ? GEOAC_EQ_3D : (GEOAC_CLI_SET == 2) ? GEOAC_EQ_GLOBAL
                     : (GEOAC_CLI_SET == 3) ? GEOAC_EQ_3D_RNGDEP : GEOAC_EQ_GLOBAL_RNGDEP;
static const bool kRng = (GEOAC_CLI_SET >= 3);                    // grid of profiles: -prop prefix loc_x loc_y, parameters from argv[5]
static const bool kRngC = (GEOAC_CLI_SET == 3), kRngS = (GEOAC_CLI_SET == 4);        // Cartesian / spherical grid main
static const bool kCart3 = (GEOAC_CLI_SET == 1 || GEOAC_CLI_SET == 3);   // x, y, z file layouts
static const bool kSph = (GEOAC_CLI_SET == 2 || GEOAC_CLI_SET == 4);     // z, lat, lon file layouts; source given as lat/lon

static bool string2bool(const string& v){          // GeoAc.Interface.cpp:125-128
    return !v.empty() && (strcasecmp(v.c_str(), "true") == 0 || atoi(v.c_str()) != 0);
}

// ---- arguments of this GPU build (not in the reference's grammar; its parsers skip what they do not know, and so do ours for the rest):
//   gpu_devices=0,1,...      HIP devices an arrivals-only -prop fan is dealt over (default: device 0)
//   gpu_stats=<file>         run summary as JSON (rays, RK4 ray-steps, GPU seconds, per-device shares)
//   gpu_rays_per_batch=<n>   rays per azimuth group of a -prop run that keeps raypath / caustic rows (default 8192)
//   gpu_fmt_threads=<n>      threads formatting the text of a -prop run (default: the host's cores, at most 16; 1: the single-thread path)
//   gpu_opt=<KEY>:<value>    a launch-plan option of the library (geoac_set_option), repeatable
// The environment (GEOAC_DEVICES, GEOAC_STATS, GEOAC_CLI_RAYS_PER_BATCH) is honoured only with GEOAC_DEBUG_ENV=1.
static string g_devices, g_stats;
static long g_rays_per_batch = 0;
static int g_fmt_threads = 0;            // gpu_fmt_threads=<n>: threads that format a -prop run's text (default: the host's cores, at most 16)
static uint64_t g_text_bytes = 0;        // of the last -prop run: bytes of text written, seconds spent formatting + writing them (gpu_stats)
static double g_text_seconds = 0.0;
static double g_fan_seconds = 0.0, g_fetch_seconds = 0.0, g_wait_seconds = 0.0;     // ... seconds in geoac_fan_run, in fetching the sample rows, waiting for the writer of the group before

// a double as an iostream in its default float format prints it at precision `prec`: `%.{prec}g` (std::num_put, C locale).  std::to_chars with
// chars_format::general and a precision is specified as exactly that conversion; -format_selftest compares the two on a few million values.
static inline int fmt_g(char* buf, double v, int prec){
    auto r = std::to_chars(buf, buf + 48, v, std::chars_format::general, prec);
    return (int)(r.ptr - buf);
}
static int format_selftest(long n){
    // every power of ten and its neighbours, the precision boundaries, specials, then pseudo-random bit patterns and magnitudes
    vector<double> vals = { 0.0, -0.0, 1.0, -1.0, 0.1, 1e-5, 9.9999995e-5, 1e-4, 99999.95, 999999.5, 1e6, 1e21, 1e-300, 1e300, 123456789.0, 0.000123456789,
                            5e-324, 1.7976931348623157e308, INFINITY, -INFINITY, NAN, 9.5, 99.5, 0.5, 2.5, 1.5e-7, 12345.678901234, 30.213245250789637, -179.99999999 };
    for(int e = -30; e <= 30; e++){ const double p = pow(10.0, e); vals.push_back(p); vals.push_back(nextafter(p, 0.0)); vals.push_back(nextafter(p, INFINITY)); vals.push_back(-9.999999 * p); vals.push_back(9.9999995 * p); }
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for(long i = 0; i < n; i++){
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        double v;
        if(i & 1){ memcpy(&v, &x, 8); }                                   // any bit pattern
        else { v = ((double)(x >> 11) / 9007199254740992.0 - 0.5) * pow(10.0, (double)((x >> 3) % 25) - 12.0); }   // magnitudes the files hold
        vals.push_back(v);
    }
    long bad = 0;
    char buf[64];
    for(double v : vals) for(int prec : {6, 8}){
        ostringstream os; os << setprecision(prec) << v;
        const int len = fmt_g(buf, v, prec);
        if(os.str() != string(buf, (size_t)len)){ if(bad < 10) cout << "MISMATCH precision " << prec << ": iostream " << os.str() << " to_chars " << string(buf, (size_t)len) << '\n'; bad++; }
    }
    cout << kName << ": -format_selftest: " << vals.size() * 2 << " conversions, " << bad << " mismatches" << '\n';
    return bad ? 1 : 0;
}
static vector<std::pair<string, string>> g_opts;
static void parse_gpu_args(int argc, char* argv[]){
    const char* dbg = getenv("GEOAC_DEBUG_ENV");
    if(dbg && atoi(dbg) != 0){
        if(const char* e = getenv("GEOAC_DEVICES")) g_devices = e;
        if(const char* e = getenv("GEOAC_STATS")) g_stats = e;
        if(const char* e = getenv("GEOAC_CLI_RAYS_PER_BATCH")) g_rays_per_batch = atol(e);
    }
    for(int i = 2; i < argc; i++){
        const char* a = argv[i];
        if(strncmp(a, "gpu_devices=", 12) == 0) g_devices = a + 12;
        else if(strncmp(a, "gpu_stats=", 10) == 0) g_stats = a + 10;
        else if(strncmp(a, "gpu_rays_per_batch=", 19) == 0){
            char* end = nullptr; const long v = strtol(a + 19, &end, 10);
            if(end == a + 19 || *end || v <= 0) cerr << kName << ": warning: " << a << ": expected a ray count > 0 - argument ignored" << '\n';
            else g_rays_per_batch = v;
        }
        else if(strncmp(a, "gpu_fmt_threads=", 16) == 0){
            char* end = nullptr; const long v = strtol(a + 16, &end, 10);
            if(end == a + 16 || *end || v <= 0 || v > 256) cerr << kName << ": warning: " << a << ": expected a thread count in 1..256 - argument ignored" << '\n';
            else g_fmt_threads = (int)v;
        }
        else if(strncmp(a, "gpu_opt=", 8) == 0){
            const char* c = strchr(a + 8, ':');
            if(c && c != a + 8 && c[1]) g_opts.emplace_back(string(a + 8, c), string(c + 1));
            else cerr << kName << ": warning: " << a << ": expected gpu_opt=<KEY>:<value> - argument ignored" << '\n';
        }
        // (the reference's parsers skip key=value pairs they do not know, and so do the loops below; an argument that LOOKS like one of this
        //  build's own and is not - a typo such as gpu_device=1 - would change nothing without a word)
        else if(strncmp(a, "gpu_", 4) == 0) cerr << kName << ": warning: unknown argument " << a << " (gpu_devices=, gpu_stats=, gpu_rays_per_batch=, gpu_fmt_threads=, gpu_opt=<KEY>:<value>) - ignored" << '\n';
    }
}
static int apply_gpu_opts(geoac_ctx* ctx){
    for(const auto& kv : g_opts){
        int rc = geoac_set_option(ctx, kv.first.c_str(), kv.second.c_str());
        if(rc){ cout << kName << ": gpu_opt=" << kv.first << ": " << geoac_last_error(ctx) << '\n'; return rc; }
    }
    return 0;
}

static vector<int> device_list(){
    vector<int> d;
    const char* e = g_devices.empty() ? nullptr : g_devices.c_str();
    if(e){
        const char* q = e;
        while(*q){
            char* end = nullptr;
            long v = strtol(q, &end, 10);
            if(end == q) break;
            if(v >= 0) d.push_back((int)v);
            q = (*end == ',') ? end + 1 : end;
            if(*end != ',' ) break;
        }
    }
    if(d.empty()) d.push_back(0);
    return d;
}

static void write_stats(const char* mode, long rays, uint64_t steps, double seconds, const vector<int>& devs,
                        const vector<uint64_t>& d_rays, const vector<uint64_t>& d_steps, const vector<uint64_t>& d_groups){
    const char* path = g_stats.c_str();
    if(!*path) return;
    ofstream js(path);
    js << "{\"program\": \"" << kName << "\", \"mode\": \"" << mode << "\", \"rays\": " << rays << ", \"rk4_ray_steps\": " << steps
       << ", \"gpu_seconds\": " << setprecision(9) << seconds << ", \"ray_steps_per_s\": " << (seconds > 0 ? steps / seconds : 0.0)
       << ", \"text_bytes\": " << g_text_bytes << ", \"text_seconds\": " << g_text_seconds << ", \"text_MB_per_s\": " << (g_text_seconds > 0 ? g_text_bytes / g_text_seconds / 1e6 : 0.0)
       << ", \"fan_run_seconds\": " << g_fan_seconds << ", \"sample_fetch_seconds\": " << g_fetch_seconds << ", \"writer_wait_seconds\": " << g_wait_seconds
       << ", \"devices\": [";
    for(size_t i = 0; i < devs.size(); i++) js << (i ? ", " : "") << devs[i];
    js << "], \"per_device\": [";
    for(size_t i = 0; i < d_rays.size(); i++)
        js << (i ? ", " : "") << "{\"rays\": " << d_rays[i] << ", \"rk4_ray_steps\": " << d_steps[i] << ", \"groups\": " << d_groups[i] << "}";
    js << "]}" << '\n';
}

static void usage(){
    cout << '\n' << "Usage: " << kName << " [option] " << (kRngS ? "profile_prefix loc_lat.dat loc_lon.dat [parameter=value ...]" : kRng ? "profile_prefix loc_x.dat loc_y.dat [parameter=value ...]" : "profile.met [parameter=value ...]") << '\n'
         << "  GPU (MI355X) build of " << kName << "; options, parameters, defaults and output files follow" << '\n'
         << "  LANL-Seismoacoustics/GeoAc (see GeoAc_Manual.pdf): -prop, -interactive" << (kEq != GEOAC_EQ_2D ? ", -eig_search, -eig_direct" : "") << '.' << '\n' << '\n';
}

// host evaluation of the profile splines for atmo.dat only (set-up / reporting code, not the ray path)
struct Profile {
    int n = 0;
    vector<double> x, T, u, v, rho, sl;
    double eval(const vector<double>& f, const double* s, double xq) const {
        if(xq > x[n - 1]) xq = x[n - 1];
        if(xq < x[0]) xq = x[0];
        int k = 0;
        while(k < n - 2 && xq > x[k + 1]) k++;
        double h = x[k + 1] - x[k], X = (xq - x[k]) / h, df = f[k + 1] - f[k];
        double A = s[k] * h - df, B = -s[k + 1] * h + df;
        return (1.0 - X) * f[k] + X * f[k + 1] + X * (1.0 - X) * (A * (1.0 - X) + B * X);
    }
    double c(double xq) const { return sqrt(kGamR * eval(T, &sl[0], xq)); }
    double uu(double xq) const { return eval(u, &sl[n], xq); }
    double vv(double xq) const { return eval(v, &sl[2 * (size_t)n], xq); }
    double rr(double xq) const { return eval(rho, &sl[3 * (size_t)n], xq); }
};

// GeoAc_WriteProfile(file, azimuth): Interface.cpp:78-98 / Interface.Global.cpp:78-97
static void write_profile(const Profile& p, const char* file_name, double azimuth){
    ofstream file_out; file_out.open(file_name);
    if(!file_out.is_open()){ cout << "Error opening file, check file name." << '\n'; return; }
    for(int m = 0; m < 1400; m++){
        double x0 = (kSph ? 6370.0 : 0.0) + m / 10.0;
        double c = p.c(x0), u = p.uu(x0), v = p.vv(x0), rho = p.rr(x0);
        file_out << x0 << '\t';
        file_out << pow(c * 1000.0, 2) / (kR * kGam) << '\t';
        file_out << u * 1000.0 << '\t';
        file_out << v * 1000.0 << '\t';
        file_out << rho << '\t';
        file_out << rho * pow(c * 1000.0, 2) / kGam * 10.0 << '\t';
        file_out << c << '\t';
        file_out << c + cos(azimuth * Pi / 180.0) * u + sin(azimuth * Pi / 180.0) * v;
        if(!kSph) file_out << '\t';            // the Cartesian writer ends the row with a tab (Interface.cpp:93-94)
        file_out << '\n';
    }
    file_out.close();
}

// GeoAc_WriteProfile(file, x0, y0, azimuth): Interface.cpp:100-121 (range-dependent atmosphere sampled above the source)
struct Grid {
    int nx = 0, ny = 0, nz = 0;
    vector<double> x, y, z, T, u, v, rho, tab;
    double f(int field, double xq, double yq, double zq) const {
        return geoac_grid_eval_eq(kEq, nx, ny, nz, x.data(), y.data(), z.data(), tab.data(), field, xq, yq, zq);
    }
};
static void write_profile_grid(const Grid& g, const char* file_name, double x0, double y0, double azimuth){
    ofstream file_out; file_out.open(file_name);
    if(!file_out.is_open()){ cout << "Error opening file, check file name." << '\n'; return; }
    for(int m = 0; m < 1400; m++){
        // Cartesian: (x_src, y_src, z) (Interface.cpp:100-121); spherical: (lat_src, lon_src [rad], r_earth + z), no trailing tab (Interface.Global.cpp:99-118)
        double z0 = (kSph ? 6370.0 : 0.0) + m / 10.0;
        double c = sqrt(kGamR * g.f(0, x0, y0, z0)), u = g.f(1, x0, y0, z0), v = g.f(2, x0, y0, z0), rho = g.f(3, x0, y0, z0);
        file_out << z0 << '\t';
        file_out << pow(c * 1000.0, 2) / (kR * kGam) << '\t';
        file_out << u * 1000.0 << '\t';
        file_out << v * 1000.0 << '\t';
        file_out << rho << '\t';
        file_out << rho * pow(c * 1000.0, 2) / kGam * 10.0 << '\t';
        file_out << c << '\t';
        file_out << c + cos(azimuth * Pi / 180.0) * u + sin(azimuth * Pi / 180.0) * v;
        if(!kSph) file_out << '\t';
        file_out << '\n';
    }
    file_out.close();
}

static int run_prop(char* inputs[], int count){
    double theta_min = 0.5, theta_max = 45.0, theta_step = 0.5;
    double phi_min = -90.0, phi_max = -90.0, phi_step = 1.0;
    int bounces = 2;
    double src_a = 0.0, src_b = 0.0, z_src = 0.0;                 // Global: lat_src, lon_src; (3D x_src, y_src are not settable in -prop)
    if(kEq == GEOAC_EQ_GLOBAL){ src_a = 30.0; src_b = 0.0; }
    bool CalcAmp = true, WriteAtmo = false, WriteRays = true, WriteCaustics = false;
    double freq = 0.1;
    const char* ProfileFormat = "zTuvdp";
    double z_grnd = 0.0, tweak_abs = 0.3;
    char input_check;
    const int arg0 = kRng ? 5 : 3;

    for(int i = arg0; i < count; i++) if(strncmp(inputs[i], "profile_format=", 15) == 0) ProfileFormat = inputs[i] + 15;

    // ---- stratified mains: load the profile BEFORE parsing the rest (z_grnd= therefore never reaches the wind taper: Q9);
    //      the range-dependent main parses first and loads with the parsed z_grnd (GeoAc3D.RngDep_main.cpp:131-170) ----
    Profile prof;
    Grid grid;
    auto load_grid = [&](double z_taper) -> int {
        if(geoac_grid_dims(inputs[2], inputs[3], inputs[4], &grid.nx, &grid.ny, &grid.nz)){ cout << "Error opening file, check file name" << '\n'; return 1; }
        const size_t nn = (size_t)grid.nx * grid.ny * grid.nz;
        grid.x.resize(grid.nx); grid.y.resize(grid.ny); grid.z.resize(grid.nz);
        grid.T.resize(nn); grid.u.resize(nn); grid.v.resize(nn); grid.rho.resize(nn);
        int lrc = geoac_grid_load_eq(kEq, inputs[2], inputs[3], inputs[4], ProfileFormat, z_taper, grid.nx, grid.ny, grid.nz, grid.x.data(), grid.y.data(), grid.z.data(),
                                     grid.T.data(), grid.u.data(), grid.v.data(), grid.rho.data());
        if(lrc == -2){ cout << "Unrecognized profile option: " << ProfileFormat << ".  Valid options are: zTuvdp and zuvwTdp" << '\n'; return 1; }
        if(lrc){ cout << "Error opening file, check file name" << '\n'; return 1; }
        return 0;
    };
    if(kRngS){
        // the spherical grid main loads first, with z_grnd = 0 (GeoAcGlobal.RngDep_main.cpp:130-133), and puts the default source at the
        // centre of the grid (:135-137)
        if(load_grid(0.0)) return 1;
        src_a = (grid.x[0] + grid.x[(size_t)grid.nx - 1]) / 2.0 * 180.0 / Pi;
        src_b = (grid.y[0] + grid.y[(size_t)grid.ny - 1]) / 2.0 * 180.0 / Pi;
    }
    if(!kRng){
    prof.n = geoac_met_rows(inputs[2]);
    if(prof.n < 3){ cout << "Error opening file, check file name" << '\n'; return 1; }
    prof.x.resize(prof.n); prof.T.resize(prof.n); prof.u.resize(prof.n); prof.v.resize(prof.n); prof.rho.resize(prof.n);
    prof.sl.resize(4 * (size_t)prof.n);
    if(geoac_met_load(inputs[2], ProfileFormat, kEq, prof.n, prof.x.data(), prof.T.data(), prof.u.data(), prof.v.data(), prof.rho.data()) != prof.n){
        cout << "Unrecognized profile option: " << ProfileFormat << ".  Valid options are: zTuvdp and zuvwTdp" << '\n';
        return 1;
    }
    geoac_natural_spline_slopes(prof.n, prof.x.data(), prof.T.data(),   &prof.sl[0]);
    geoac_natural_spline_slopes(prof.n, prof.x.data(), prof.u.data(),   &prof.sl[prof.n]);
    geoac_natural_spline_slopes(prof.n, prof.x.data(), prof.v.data(),   &prof.sl[2 * (size_t)prof.n]);
    geoac_natural_spline_slopes(prof.n, prof.x.data(), prof.rho.data(), &prof.sl[3 * (size_t)prof.n]);
    }

    geoac_params P;
    geoac_default_params(kEq, &P);
    if(!kRng) P.vert_limit = prof.x[prof.n - 1];                   // GeoAc_SetPropRegion

    for(int i = arg0; i < count; i++){
        const char* a = inputs[i];
        if(strncmp(a, "theta_min=", 10) == 0){ theta_min = atof(a + 10); }
        else if(strncmp(a, "theta_max=", 10) == 0){ theta_max = atof(a + 10); }
        else if(strncmp(a, "theta_step=", 11) == 0){ theta_step = atof(a + 11); }
        else if(kEq != GEOAC_EQ_2D && strncmp(a, "phi_min=", 8) == 0){ phi_min = atof(a + 8); }
        else if(kEq != GEOAC_EQ_2D && strncmp(a, "phi_max=", 8) == 0){ phi_max = atof(a + 8); }
        else if(kEq != GEOAC_EQ_2D && strncmp(a, "phi_step=", 9) == 0){ phi_step = atof(a + 9); }
        else if(strncmp(a, "azimuth=", 8) == 0){ phi_min = atof(a + 8); phi_max = atof(a + 8); phi_step = 1.0; }
        else if(strncmp(a, "bounces=", 8) == 0){ bounces = atoi(a + 8); }
        else if(kSph && strncmp(a, "lat_src=", 8) == 0){ src_a = atof(a + 8); }
        else if(kSph && strncmp(a, "lon_src=", 8) == 0){ src_b = atof(a + 8); }
        else if(kRngC && strncmp(a, "x_src=", 6) == 0){ src_a = atof(a + 6); }
        else if(kRngC && strncmp(a, "y_src=", 6) == 0){ src_b = atof(a + 6); }
        // the range-dependent main takes the region limits but GeoAc_SetPropRegion overwrites them right after loading (Q9)
        else if(kRngC && (strncmp(a, "x_min=", 6) == 0 || strncmp(a, "x_max=", 6) == 0 || strncmp(a, "y_min=", 6) == 0 || strncmp(a, "y_max=", 6) == 0)){ }
        else if(kRngC && strncmp(a, "alt_max=", 8) == 0){ }
        // spherical grid main: the box limits are assigned as typed and compared with radians (GeoAcGlobal.RngDep_main.cpp:166-169)
        else if(kRngS && strncmp(a, "lat_min=", 8) == 0){ P.xy_limits[0] = atof(a + 8); }
        else if(kRngS && strncmp(a, "lat_max=", 8) == 0){ P.xy_limits[1] = atof(a + 8); }
        else if(kRngS && strncmp(a, "lon_min=", 8) == 0){ P.xy_limits[2] = atof(a + 8); }
        else if(kRngS && strncmp(a, "lon_max=", 8) == 0){ P.xy_limits[3] = atof(a + 8); }
        else if(strncmp(a, "z_src=", 6) == 0){ z_src = atof(a + 6); }
        else if(strncmp(a, "z_grnd=", 7) == 0){ z_grnd = atof(a + 7); }
        else if(kEq != GEOAC_EQ_2D && strncmp(a, "WriteAtmo=", 10) == 0){ WriteAtmo = string2bool(a + 10); }
        else if(kEq != GEOAC_EQ_2D && strncmp(a, "WriteRays=", 10) == 0){ WriteRays = string2bool(a + 10); }
        else if(strncmp(a, "freq=", 5) == 0){ freq = atof(a + 5); }
        else if(strncmp(a, "abs_coeff=", 10) == 0){ tweak_abs = max(0.0, atof(a + 10)); }
        else if(strncmp(a, "profile_format=", 15) == 0){ ProfileFormat = a + 15; }
        else if(strncmp(a, "WriteCaustics=", 14) == 0){ WriteCaustics = string2bool(a + 14); }
        else if(strncmp(a, "CalcAmp=", 8) == 0){ CalcAmp = string2bool(a + 8); }
        else if(strncmp(a, "alt_max=", 8) == 0){ P.vert_limit = atof(a + 8); }     // Global: a km altitude compared with a radius (Q9)
        else if(!kRng && strncmp(a, "rng_max=", 8) == 0){ P.range_limit = atof(a + 8); }
        else if(strncmp(a, "gpu_", 4) == 0){ }                        // arguments of this GPU build (parse_gpu_args)
        else {
            cout << "***WARNING*** Unrecognized parameter entry: " << a << '\n';
            cout << "Continue? (y/n):"; cin >> input_check;
            if(input_check != 'y' && input_check != 'Y') return 0;
        }
    }
    if(kEq == GEOAC_EQ_2D){ WriteRays = true; WriteAtmo = true; }     // GeoAc2D always writes raypaths and atmo.dat
    if(WriteCaustics) CalcAmp = true;
    if(kRngC){
        z_src = max(z_grnd, z_src);                                 // GeoAc3D.RngDep_main.cpp:165
        if(load_grid(z_grnd)) return 1;                              // this main parses first and loads with the parsed z_grnd (:131-170)
    }

    // output prefix = input path up to the first '.' (GeoAcGlobal_main.cpp:170-177)
    char file_title[64];
    { int m = 0; for(; m < 50 && inputs[2][m] != '\0' && inputs[2][m] != '.'; m++) file_title[m] = inputs[2][m]; file_title[m] = '\0'; }
    char output_buffer[96];

    if(WriteAtmo && !kRng) write_profile(prof, "atmo.dat", 90.0 - phi_min);
    if(WriteAtmo && kRng){
        grid.tab.resize(geoac_grid_table_size(grid.nx, grid.ny, grid.nz));
        geoac_grid_table_eq(kEq, grid.nx, grid.ny, grid.nz, grid.x.data(), grid.y.data(), grid.z.data(), grid.T.data(), grid.u.data(), grid.v.data(), grid.rho.data(), grid.tab.data());
        if(kRngS) write_profile_grid(grid, "atmo.dat", src_a * Pi / 180.0, src_b * Pi / 180.0, 90.0 - phi_min);
        else      write_profile_grid(grid, "atmo.dat", src_a, src_b, 90.0 - phi_min);
    }

    // ---- the fan on the GPU(s) ----
    const vector<int> devs = device_list();
    const bool multi = devs.size() > 1 && !(WriteRays || WriteCaustics);      // several devices: arrivals-only fans (sample capture is per context)
    geoac_ctx* ctx = nullptr;
    geoac_pool* pool = nullptr;
    int rc;
    if(multi){
        rc = geoac_pool_create(&pool, kEq, (int)devs.size(), devs.data());
        if(rc){ cout << kName << ": " << geoac_strerror(rc) << '\n'; return 2; }
        for(int d = 0; d < geoac_pool_size(pool); d++) if(apply_gpu_opts(geoac_pool_ctx(pool, d))) return 2;
        if(kRng) rc = geoac_pool_upload_atmo_3d(pool, grid.nx, grid.ny, grid.nz, grid.x.data(), grid.y.data(), grid.z.data(), grid.T.data(), grid.u.data(), grid.v.data(), grid.rho.data());
        else     rc = geoac_pool_upload_atmo_1d(pool, prof.n, prof.x.data(), prof.T.data(), prof.u.data(), prof.v.data(), prof.rho.data(), prof.sl.data());
        if(rc){ cout << kName << ": " << geoac_pool_last_error(pool) << '\n'; return 2; }
        ctx = geoac_pool_ctx(pool, 0);
    } else {
    rc = geoac_create(&ctx, kEq, devs[0]);
    if(rc){ cout << kName << ": " << geoac_strerror(rc) << '\n'; return 2; }
    if(apply_gpu_opts(ctx)) return 2;
    if(kRng) rc = geoac_upload_atmo_3d(ctx, grid.nx, grid.ny, grid.nz, grid.x.data(), grid.y.data(), grid.z.data(), grid.T.data(), grid.u.data(), grid.v.data(), grid.rho.data());
    else     rc = geoac_upload_atmo_1d(ctx, prof.n, prof.x.data(), prof.T.data(), prof.u.data(), prof.v.data(), prof.rho.data(), prof.sl.data());
    if(rc){ cout << kName << ": " << geoac_last_error(ctx) << '\n'; return 2; }
    }
    P.z_grnd = z_grnd; P.tweak_abs = tweak_abs; P.freq = freq; P.bounces = bounces; P.calc_amp = CalcAmp ? 1 : 0;
    P.mode = (WriteRays ? GEOAC_MODE_WRITE_RAYS : 0) | (WriteCaustics ? GEOAC_MODE_WRITE_CAUSTICS : 0);
    if(kSph){ P.src[0] = z_src; P.src[1] = src_a; P.src[2] = src_b; }
    else if(kEq == GEOAC_EQ_3D){ P.src[0] = 0.0; P.src[1] = 0.0; P.src[2] = z_src; }
    else if(kRngC){ P.src[0] = src_a; P.src[1] = src_b; P.src[2] = z_src; }
    else { P.src[0] = z_src; P.src[1] = 0.0; P.src[2] = 0.0; }
    rc = multi ? geoac_pool_set_params(pool, &P) : geoac_set_params(ctx, &P);
    if(rc){ cout << kName << ": " << (multi ? geoac_pool_last_error(pool) : geoac_last_error(ctx)) << '\n'; return 2; }

    long nr;
    if(kEq == GEOAC_EQ_2D) nr = geoac_fan_enumerate(theta_min, theta_max, theta_step, phi_min, phi_min, 1.0, 0, nullptr, nullptr);
    else nr = geoac_fan_enumerate(theta_min, theta_max, theta_step, phi_min, phi_max, phi_step, 0, nullptr, nullptr);
    if(nr < 0) nr = 0;
    vector<double> th((size_t)max(nr, 1L)), ph((size_t)max(nr, 1L));
    if(kEq == GEOAC_EQ_2D) geoac_fan_enumerate(theta_min, theta_max, theta_step, phi_min, phi_min, 1.0, nr, th.data(), ph.data());
    else geoac_fan_enumerate(theta_min, theta_max, theta_step, phi_min, phi_max, phi_step, nr, th.data(), ph.data());
    const int legs = bounces + 1;
    // ---- files, in the reference's formats ----
    ofstream results, raypath;
    sprintf(output_buffer, "%s_results.dat", file_title);
    results.open(output_buffer);
    if(kSph){
        results << "# theta [deg]" << '\t' << "phi [deg]" << '\t' << (kRngS ? "Bounces" : "n_b") << '\t' << "lat_0 [deg]" << '\t' << "lon_0 [deg]" << '\t' << "Travel Time [s]"
                << '\t' << "Celerity [km/s]" << '\t' << "Turning Height [km]" << '\t' << "Inclination [deg]" << '\t' << "Back Azimuth [deg]"
                << '\t' << "Geo. Atten. [dB]" << '\t' << "Atmo. Atten. [dB]" << '\n';
    } else if(kCart3){
        results << "# theta [deg]" << '\t' << "phi [deg]" << '\t' << "n_b" << '\t' << "x_0 [km]" << '\t' << "y_0 [km]" << '\t' << "Travel Time [s]"
                << '\t' << "Turning Height [km]" << '\t' << "Inclination [deg]" << '\t' << "Back Azimuth [deg]"
                << '\t' << "Geo. Atten. [dB]" << '\t' << "Atmo. Atten. [dB]" << '\n';
    } else {
        results << "# theta [deg]" << '\t' << "phi [deg]" << '\t' << "n_b" << '\t' << "r_0 [km]" << '\t' << "Travel Time [s]"
                << '\t' << "Turning Height [km]" << '\t' << "Inclination [deg]" << '\t' << "Geo. Atten. [dB]" << '\t' << "Atmo. Atten. [dB]" << '\n';
    }
    if(WriteRays){
        sprintf(output_buffer, "%s_raypaths.dat", file_title);
        raypath.open(output_buffer);
        if(kSph)                    raypath << "# z [km]" << '\t' << "Lat [deg]" << '\t' << "Long [deg]";
        else if(kCart3)             raypath << "# x [km]" << '\t' << "y [km]" << '\t' << "z [km]";
        else                        raypath << "# r [km]" << '\t' << "z [km]";
        raypath << '\t' << "Geo. Atten. [dB]" << '\t' << "Atmo. Atten. [dB]" << '\t' << "Travel Time [s]" << '\n';
    }
    vector<ofstream> caustics;
    if(WriteCaustics){
        caustics.resize((size_t)legs);
        for(int bnc = 0; bnc <= bounces; bnc++){
            sprintf(output_buffer, "%s_caustics-path%i.dat", file_title, bnc);
            caustics[(size_t)bnc].open(output_buffer);
            if(kSph)                    caustics[(size_t)bnc] << "# z [km]" << '\t' << "Lat [deg]" << '\t' << "Long [deg]";
            else if(kCart3)             caustics[(size_t)bnc] << "# x [km]" << '\t' << "y [km]" << '\t' << "z [km]";
            else                        caustics[(size_t)bnc] << "# r [km]" << '\t' << "z [km]";
            caustics[(size_t)bnc] << '\t' << "Travel Time [s]" << '\n';
        }
    }

    // ---- the text of the files.  The reference's loop (GeoAcGlobal_main.cpp:256-317 and twins) prints through iostreams in their default float format,
    //      `%.{precision}g`; every stream starts at precision 6 and the spherical mains switch a stream to 8 at its first latitude field, where it
    //      STICKS (Q14): whatever a stream prints before its first setprecision(8) - the height of the first raypath / caustic row, theta and phi of the first
    //      results row - has 6 digits, everything after 8.  Formatting is the whole cost of a WriteRays run (1.1 GB of text for 16 200 rays against
    //      0.1 s of GPU time), so a batch is cut into chunks of rays that are formatted on several threads into buffers - std::to_chars, the same
    //      `%.{p}g` digits as the streams' (self-test: -format_selftest) - and written in order.  A chunk cannot know whether an earlier one has already
    //      switched a stream, so it formats the sticky form and, of its FIRST row per stream, also the fresh form; the writer picks. ----
    struct StreamText { string text; size_t first_len = 0; string first_fresh; bool any = false; };      // text = sticky form; its first row is text[0 .. first_len)
    struct ChunkText { string progress; StreamText ray, res; vector<StreamText> cau; };
    const int p_lead_sticky = kSph ? 8 : 6, p_rest = kSph ? 8 : 6;
    auto format_chunk = [&](long i0, long i1, long base, const vector<double>& rec, const vector<double>& smp, const vector<size_t>& smp_of_ray, ChunkText& out){
        out.cau.resize(WriteCaustics ? (size_t)legs : 0);
        char tmp[64];
        auto num = [&](string& d, double v, int prec){ d.append(tmp, (size_t)fmt_g(tmp, v, prec)); };
        // one row into stream st: `lead` writes the fields printed before the row's first setprecision(8) at precision p, `rest` the others
        auto row = [&](StreamText& st, auto&& lead, auto&& rest){
            const size_t at = st.text.size();
            lead(st.text, p_lead_sticky); rest(st.text);
            if(!st.any){
                st.any = true; st.first_len = st.text.size() - at;            // (at == 0)
                lead(st.first_fresh, 6); rest(st.first_fresh);
            }
        };
        for(long i = i0; i < i1; i++){
            out.progress += "Plotting ray path w/ theta = "; num(out.progress, th[(size_t)i], 6); out.progress += ", phi = "; num(out.progress, ph[(size_t)i], 6);
            out.progress += kRngC ? ".\n" : "\n";
            // raypath / caustic rows of this ray (sorted by ray, leg, m)
            for(size_t sp = smp_of_ray[(size_t)(i - base)]; sp < smp_of_ray[(size_t)(i - base) + 1]; sp++){
                const double* S = &smp[sp * GEOAC_SMP_STRIDE];
                const int leg = (int)S[GEOAC_SMP_LEG], kind = (int)S[GEOAC_SMP_KIND];
                const double* v = S + GEOAC_SMP_V0;
                if(kind == 0 && WriteRays){
                    const int nv = (kSph || kCart3) ? 6 : 5;
                    row(out.ray, [&](string& d, int p){ num(d, v[0], p); },
                                 [&](string& d){ for(int q = 1; q < nv; q++){ d += '\t'; num(d, v[q], p_rest); } d += '\n'; });
                } else if(kind == 1 && WriteCaustics && leg < legs){
                    // (the range-dependent Cartesian main writes a literal 0.0 column before the time, :270-274; the sample row carries it)
                    const int nv = kSph ? 4 : (kRngC ? 5 : (kEq == GEOAC_EQ_3D ? 4 : 3));
                    row(out.cau[(size_t)leg], [&](string& d, int p){ num(d, v[0], p); },
                                              [&](string& d){ for(int q = 1; q < nv; q++){ d += '\t'; num(d, v[q], p_rest); } d += '\n'; });
                }
            }
            for(int b = 0; b < legs; b++){
                const double* R = &rec[((size_t)(i - base) * legs + b) * GEOAC_REC_STRIDE];
                if(R[GEOAC_REC_VALID] == 0.0) break;                    // BreakCheck: no row for this and later legs
                row(out.res, [&](string& d, int p){ num(d, th[(size_t)i], p); d += '\t'; num(d, ph[(size_t)i], p); d += '\t'; d += to_string(b); },
                    [&](string& d){
                        auto f = [&](double x){ d += '\t'; num(d, x, p_rest); };
                        if(kSph){
                            f(R[GEOAC_REC_STATE + 1] * 180.0 / Pi); f(R[GEOAC_REC_STATE + 2] * 180.0 / Pi); f(R[GEOAC_REC_TTIME]); f(R[GEOAC_REC_RANGE] / R[GEOAC_REC_TTIME]);
                            f(R[GEOAC_REC_TURN]); f(R[GEOAC_REC_INCL]); f(R[GEOAC_REC_BACKAZ]);
                        } else if(kCart3){
                            f(R[GEOAC_REC_STATE + 0]); f(R[GEOAC_REC_STATE + 1]); f(R[GEOAC_REC_TTIME]); f(R[GEOAC_REC_TURN]); f(R[GEOAC_REC_INCL]); f(R[GEOAC_REC_BACKAZ]);
                        } else {
                            f(R[GEOAC_REC_STATE + 0]); f(R[GEOAC_REC_TTIME]); f(R[GEOAC_REC_TURN]); f(R[GEOAC_REC_INCL]);
                        }
                        f(CalcAmp ? 20.0 * log10(R[GEOAC_REC_AMP]) : 0.0);
                        f(-R[GEOAC_REC_ATTEN]);
                        d += '\n';
                    });
            }
            if(WriteRays) out.ray.text += '\n';                          // blank line after each ray
            // blank line after each azimuth in results (not in GeoAc2D)
            if(kEq != GEOAC_EQ_2D && (i + 1 == nr || ph[(size_t)i + 1] != ph[(size_t)i])) out.res.text += '\n';
        }
    };
    // stream state across the whole run: has the stream seen its first setprecision(8) (spherical mains: its first row)?
    bool ray_switched = false, res_switched = false;
    vector<char> cau_switched((size_t)legs, 0);
    uint64_t text_bytes = 0;
    double text_seconds = 0.0;
    auto emit = [&](ostream& os, const StreamText& st, bool& switched){
        if(st.any && kSph && !switched){
            os.write(st.first_fresh.data(), (streamsize)st.first_fresh.size());
            os.write(st.text.data() + st.first_len, (streamsize)(st.text.size() - st.first_len));
            text_bytes += st.first_fresh.size() + st.text.size() - st.first_len;
            switched = true;
        } else {
            os.write(st.text.data(), (streamsize)st.text.size());
            text_bytes += st.text.size();
        }
    };
    const int n_fmt_threads = max(1, g_fmt_threads > 0 ? g_fmt_threads : (int)min(16u, max(1u, std::thread::hardware_concurrency())));
    // one batch of rays [i0, i1) of the fan: rec = its records, smp = its sample rows (ray index relative to i0)
    auto write_batch = [&](long i0, long i1, const vector<double>& rec, const vector<double>& smp){
        const auto t0 = std::chrono::steady_clock::now();
        const size_t nsmp = smp.size() / GEOAC_SMP_STRIDE;
        vector<size_t> smp_of_ray((size_t)(i1 - i0) + 1, nsmp);           // first sample row of every ray of the batch (rows are sorted by ray)
        {
            size_t sp = 0;
            for(long i = i0; i < i1; i++){
                smp_of_ray[(size_t)(i - i0)] = sp;
                while(sp < nsmp && (long)smp[sp * GEOAC_SMP_STRIDE + GEOAC_SMP_RAY] == i - i0) sp++;
            }
            smp_of_ray[(size_t)(i1 - i0)] = sp;
        }
        const long chunk = 64;                                             // rays per chunk: ~4 MB of text with raypaths
        const long n_chunks = (i1 - i0 + chunk - 1) / chunk;
        vector<ChunkText> texts((size_t)n_chunks);
        vector<char> ready((size_t)n_chunks, 0);
        std::mutex mu; std::condition_variable cv;
        std::atomic<long> next{0};
        auto worker = [&]{
            for(;;){
                const long c = next.fetch_add(1);
                if(c >= n_chunks) return;
                format_chunk(i0 + c * chunk, min(i1, i0 + (c + 1) * chunk), i0, rec, smp, smp_of_ray, texts[(size_t)c]);
                { std::lock_guard<std::mutex> lk(mu); ready[(size_t)c] = 1; }
                cv.notify_all();
            }
        };
        vector<std::thread> pool;
        const int nt = (int)min<long>(n_fmt_threads, n_chunks);
        for(int t = 1; t < nt; t++) pool.emplace_back(worker);
        if(nt <= 1) worker();
        // in order, as soon as a chunk is ready (this thread writes; with one thread it has formatted everything above)
        for(long c = 0; c < n_chunks; c++){
            if(nt > 1){
                std::unique_lock<std::mutex> lk(mu);
                // (this thread formats too while it would otherwise wait for chunk c)
                while(!ready[(size_t)c]){
                    lk.unlock();
                    const long mine = next.fetch_add(1);
                    if(mine < n_chunks){
                        format_chunk(i0 + mine * chunk, min(i1, i0 + (mine + 1) * chunk), i0, rec, smp, smp_of_ray, texts[(size_t)mine]);
                        lk.lock(); ready[(size_t)mine] = 1; cv.notify_all();
                    } else { lk.lock(); cv.wait(lk, [&]{ return ready[(size_t)c] != 0; }); }
                }
            }
            ChunkText& T = texts[(size_t)c];
            cout.write(T.progress.data(), (streamsize)T.progress.size());
            if(WriteRays) emit(raypath, T.ray, ray_switched);
            emit(results, T.res, res_switched);
            for(size_t l = 0; l < T.cau.size(); l++){ bool sw = cau_switched[l] != 0; emit(caustics[l], T.cau[l], sw); cau_switched[l] = sw ? 1 : 0; }
            T = ChunkText();                                                // (release the text)
        }
        for(auto& t : pool) t.join();
        text_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };

    // ---- the fan on the GPU, the files on a writer thread.  Arrivals-only runs (WriteRays=False, no caustics) are ONE fan launch.
    //      Runs that keep raypath / caustic rows go azimuth group by azimuth group (about 8000 rays each), so the sample list stays
    //      bounded, and the text of group g is formatted and written while group g+1 is on the GPU (two buffers in rotation). ----
    const bool sampling = WriteRays || WriteCaustics;
    int64_t sample_cap = 16ll << 20;
    if(!multi) geoac_fan_set_sample_capacity(ctx, sample_cap);
    vector<long> az_start;                                            // first ray of every azimuth (rays are phi-major)
    for(long i = 0; i < nr; i++) if(i == 0 || ph[(size_t)i] != ph[(size_t)i - 1]) az_start.push_back(i);
    az_start.push_back(nr);
    const long n_az = (long)az_start.size() - 1;
    long az_per_batch = n_az;
    if(sampling && n_az > 0){
        const long rays_per_az = max(1L, nr / n_az);
        az_per_batch = max(1L, (g_rays_per_batch > 0 ? g_rays_per_batch : 8192L) / rays_per_az);    // (gpu_rays_per_batch=: tests force several groups on a small fan)
    }
    struct Batch { long i0 = 0, i1 = 0; vector<double> rec, smp; };
    Batch buf[2];
    std::thread writer;
    uint64_t steps = 0;
    int fail_rc = 0;
    int nb = 0;
    int capacity_retries = 0;
    const auto t_gpu0 = std::chrono::steady_clock::now();
    if(multi){
        // arrivals-only fan over the pool's devices: azimuth groups from a shared queue, records straight into the fan's table
        vector<double> rec((size_t)nr * legs * GEOAC_REC_STRIDE, 0.0), smp;
        rc = geoac_pool_fan_run(pool, (int)nr, th.data(), ph.data(), 0, rec.data(), &steps);
        if(rc){ cout << kName << ": " << geoac_pool_last_error(pool) << '\n'; geoac_pool_destroy(pool); return 2; }
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_gpu0).count();
        write_batch(0, nr, rec, smp);
        vector<uint64_t> dr(devs.size()), ds(devs.size()), dg(devs.size());
        geoac_pool_last_shares(pool, dr.data(), ds.data(), dg.data());
        g_text_bytes = text_bytes; g_text_seconds = text_seconds;
        write_stats("-prop", nr, steps, secs, devs, dr, ds, dg);
        geoac_pool_destroy(pool);
        results.close();
        cerr << kName << ": " << nr << " rays, " << steps << " RK4 ray-steps on " << devs.size() << " GPUs (";
        for(size_t i = 0; i < devs.size(); i++) cerr << (i ? ", " : "") << "device " << devs[i] << ": " << dr[i] << " rays";
        cerr << ")" << '\n';
        return 0;
    }
    for(long a0 = 0; a0 < n_az && !fail_rc; ){
        long a1 = min(n_az, a0 + az_per_batch);
        Batch& B = buf[nb & 1];
        // the buffer's previous batch (two batches ago) was handed to the writer before the last one: joined below before reuse
        const long i0 = az_start[(size_t)a0], i1 = az_start[(size_t)a1];
        vector<double> rec((size_t)(i1 - i0) * legs * GEOAC_REC_STRIDE, 0.0), smp;
        uint64_t st = 0;
        const auto t_f0 = std::chrono::steady_clock::now();
        rc = geoac_fan_run(ctx, (int)(i1 - i0), th.data() + i0, ph.data() + i0, rec.data(), &st);
        g_fan_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_f0).count();
        if(rc == GEOAC_E_CAPACITY && sampling && capacity_retries < 8){
            // GEOAC_E_CAPACITY has three causes; only a sample list that was too small is cured by splitting the group or growing the
            // list (need > current capacity).  A ray at step_limit or an overflow of the per-epoch event list come back with
            // need <= capacity: retrying those would launch the same fan again and again - report them instead.
            int64_t need = 0;
            geoac_fan_sample_count(ctx, &need);
            if(need > sample_cap){
                capacity_retries++;
                if(need > (16ll << 20) && a1 - a0 > 1){ az_per_batch = max(1L, (a1 - a0) / 2); continue; }      // split the group
                if(need <= (192ll << 20)){ sample_cap = need + need / 8; geoac_fan_set_sample_capacity(ctx, sample_cap); continue; }    // one azimuth that long: grow the list
            }
        }
        if(rc){ cout << kName << ": " << geoac_last_error(ctx) << '\n'; fail_rc = 2; break; }
        steps += st;
        int64_t ns = 0;
        geoac_fan_sample_count(ctx, &ns);
        const auto t_s0 = std::chrono::steady_clock::now();
        smp.resize((size_t)max<int64_t>(ns, 1) * GEOAC_SMP_STRIDE);
        if(ns > 0 && geoac_fan_fetch_samples(ctx, smp.data(), ns)){ cout << kName << ": " << geoac_last_error(ctx) << '\n'; fail_rc = 2; break; }
        smp.resize((size_t)ns * GEOAC_SMP_STRIDE);
        g_fetch_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_s0).count();
        const auto t_w0 = std::chrono::steady_clock::now();
        if(writer.joinable()) writer.join();                           // batches are written in order, one at a time
        g_wait_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_w0).count();
        B.i0 = i0; B.i1 = i1; B.rec.swap(rec); B.smp.swap(smp);
        writer = std::thread([&write_batch, &B]{ write_batch(B.i0, B.i1, B.rec, B.smp); });
        nb++;
        a0 = a1;
    }
    if(writer.joinable()) writer.join();
    geoac_destroy(ctx);
    if(fail_rc) return fail_rc;
    if(WriteRays) raypath.close();
    results.close();
    for(auto& c : caustics) c.close();
    g_text_bytes = text_bytes; g_text_seconds = text_seconds;
    write_stats("-prop", nr, steps, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_gpu0).count(), vector<int>(1, devs[0]),
                vector<uint64_t>(1, (uint64_t)nr), vector<uint64_t>(1, steps), vector<uint64_t>(1, (uint64_t)nb));
    cerr << kName << ": " << nr << " rays, " << steps << " RK4 ray-steps on the GPU" << '\n';
    return 0;
}

// ---- -interactive (GeoAc2D_main.cpp:239-372, GeoAc3D_main.cpp:315-456, GeoAcGlobal_main.cpp:334-492, GeoAc3D.RngDep_main.cpp:336-496,
//      GeoAcGlobal.RngDep_main.cpp:345-516): one ray per prompt through the same fan backend (a fan of one ray, WriteRays form of the
//      sums); prompts, raypath.dat / caustics.dat and the arrival summary as the reference prints them ----
static int run_interactive(char* inputs[], int count){
    double src_a = 0.0, src_b = 0.0, z_src = 0.0;
    if(kEq == GEOAC_EQ_GLOBAL){ src_a = 30.0; src_b = 0.0; }
    double freq = 0.1;
    bool CalcAmp = true, WriteCaustics = false;
    const char* ProfileFormat = "zTuvdp";
    double z_grnd = 0.0, tweak_abs = 0.3;
    char input_check;
    const int arg0 = kRng ? 5 : 3;

    Profile prof;
    Grid grid;
    auto load_grid = [&](double z_taper) -> int {
        if(geoac_grid_dims(inputs[2], inputs[3], inputs[4], &grid.nx, &grid.ny, &grid.nz)){ cout << "Error opening file, check file name" << '\n'; return 1; }
        const size_t nn = (size_t)grid.nx * grid.ny * grid.nz;
        grid.x.resize(grid.nx); grid.y.resize(grid.ny); grid.z.resize(grid.nz);
        grid.T.resize(nn); grid.u.resize(nn); grid.v.resize(nn); grid.rho.resize(nn);
        int lrc = geoac_grid_load_eq(kEq, inputs[2], inputs[3], inputs[4], ProfileFormat, z_taper, grid.nx, grid.ny, grid.nz, grid.x.data(), grid.y.data(), grid.z.data(),
                                     grid.T.data(), grid.u.data(), grid.v.data(), grid.rho.data());
        if(lrc == -2){ cout << "Unrecognized profile option: " << ProfileFormat << ".  Valid options are: zTuvdp and zuvwTdp" << '\n'; return 1; }
        if(lrc){ cout << "Error opening file, check file name" << '\n'; return 1; }
        return 0;
    };
    auto load_profile = [&](double z_taper) -> int {
        prof.n = geoac_met_rows(inputs[2]);
        if(prof.n < 3){ cout << "Error opening file, check file name" << '\n'; return 1; }
        prof.x.resize(prof.n); prof.T.resize(prof.n); prof.u.resize(prof.n); prof.v.resize(prof.n); prof.rho.resize(prof.n);
        prof.sl.resize(4 * (size_t)prof.n);
        if(geoac_met_load_zg(inputs[2], ProfileFormat, kEq, z_taper, prof.n, prof.x.data(), prof.T.data(), prof.u.data(), prof.v.data(), prof.rho.data()) != prof.n){
            cout << "Unrecognized profile option: " << ProfileFormat << ".  Valid options are: zTuvdp and zuvwTdp" << '\n';
            return 1;
        }
        geoac_natural_spline_slopes(prof.n, prof.x.data(), prof.T.data(),   &prof.sl[0]);
        geoac_natural_spline_slopes(prof.n, prof.x.data(), prof.u.data(),   &prof.sl[prof.n]);
        geoac_natural_spline_slopes(prof.n, prof.x.data(), prof.v.data(),   &prof.sl[2 * (size_t)prof.n]);
        geoac_natural_spline_slopes(prof.n, prof.x.data(), prof.rho.data(), &prof.sl[3 * (size_t)prof.n]);
        return 0;
    };

    geoac_params P;
    geoac_default_params(kEq, &P);

    if(kRngS){
        // spherical grid main: profile_format= and z_grnd= are picked up first, then the grid is loaded (taper on the parsed z_grnd) and
        // the default source put at the centre of the grid (GeoAcGlobal.RngDep_main.cpp:350-361)
        for(int i = arg0; i < count; i++){
            if(strncmp(inputs[i], "profile_format=", 15) == 0) ProfileFormat = inputs[i] + 15;
            else if(strncmp(inputs[i], "z_grnd=", 7) == 0) z_grnd = atof(inputs[i] + 7);
        }
        if(load_grid(z_grnd)) return 1;
        src_a = (grid.x[0] + grid.x[(size_t)grid.nx - 1]) / 2.0 * 180.0 / Pi;
        src_b = (grid.y[0] + grid.y[(size_t)grid.ny - 1]) / 2.0 * 180.0 / Pi;
    } else if(!kRng){
        for(int i = arg0; i < count; i++) if(strncmp(inputs[i], "profile_format=", 15) == 0) ProfileFormat = inputs[i] + 15;
        if(load_profile(0.0)) return 1;                              // loaded before z_grnd= is parsed (Q9)
        P.vert_limit = prof.x[prof.n - 1];                           // GeoAc_SetPropRegion
    }

    for(int i = arg0; i < count; i++){
        const char* a = inputs[i];
        if(kSph && strncmp(a, "lat_src=", 8) == 0){ src_a = atof(a + 8); }
        else if(kSph && strncmp(a, "lon_src=", 8) == 0){ src_b = atof(a + 8); }
        else if(kRngC && strncmp(a, "x_src=", 6) == 0){ src_a = atof(a + 6); }
        else if(kRngC && strncmp(a, "y_src=", 6) == 0){ src_b = atof(a + 6); }
        else if(strncmp(a, "z_src=", 6) == 0){ z_src = atof(a + 6); }
        else if(strncmp(a, "freq=", 5) == 0){ freq = atof(a + 5); }
        else if(strncmp(a, "abs_coeff=", 10) == 0){ tweak_abs = max(0.0, atof(a + 10)); }
        else if(strncmp(a, "z_grnd=", 7) == 0){ z_grnd = atof(a + 7); }
        else if(strncmp(a, "profile_format=", 15) == 0){ ProfileFormat = a + 15; }
        else if(strncmp(a, "WriteCaustics=", 14) == 0){ WriteCaustics = string2bool(a + 14); }
        else if(strncmp(a, "CalcAmp=", 8) == 0){ CalcAmp = string2bool(a + 8); }
        else if(strncmp(a, "alt_max=", 8) == 0){ P.vert_limit = atof(a + 8); }
        else if(!kRng && strncmp(a, "rng_max=", 8) == 0){ P.range_limit = atof(a + 8); }
        else if(kRngC && (strncmp(a, "x_min=", 6) == 0 || strncmp(a, "x_max=", 6) == 0 || strncmp(a, "y_min=", 6) == 0 || strncmp(a, "y_max=", 6) == 0)){ }
        else if(kRngS && strncmp(a, "lat_min=", 8) == 0){ P.xy_limits[0] = atof(a + 8); }
        else if(kRngS && strncmp(a, "lat_max=", 8) == 0){ P.xy_limits[1] = atof(a + 8); }
        else if(kRngS && strncmp(a, "lon_min=", 8) == 0){ P.xy_limits[2] = atof(a + 8); }
        else if(kRngS && strncmp(a, "lon_max=", 8) == 0){ P.xy_limits[3] = atof(a + 8); }
        else if(strncmp(a, "gpu_", 4) == 0){ }                        // arguments of this GPU build (parse_gpu_args)
        else {
            cout << "***WARNING*** Unrecognized parameter entry: " << a << '\n';
            cout << "Continue? (y/n):"; cin >> input_check;
            if(input_check != 'y' && input_check != 'Y') return 0;
        }
    }
    z_src = max(z_src, z_grnd);
    if(WriteCaustics) CalcAmp = true;
    if(kEq == GEOAC_EQ_GLOBAL){
        // GeoAcGlobal loads the profile a SECOND time here (GeoAcGlobal_main.cpp:367): the taper now sees the parsed z_grnd, and
        // GeoAc_SetPropRegion inside the loader puts alt_max / rng_max back to the profile top / 1500 km
        if(load_profile(z_grnd)) return 1;
        geoac_params D; geoac_default_params(kEq, &D);
        P.vert_limit = prof.x[prof.n - 1]; P.range_limit = D.range_limit;
    }
    if(kRngC){
        // parsed first, loaded after, GeoAc_SetPropRegion then overwrites alt_max and the box (GeoAc3D.RngDep_main.cpp:370-372)
        if(load_grid(z_grnd)) return 1;
        geoac_params D; geoac_default_params(kEq, &D);
        P.vert_limit = D.vert_limit;
    }
    // the range-dependent mains integrate the auxiliary equations whatever CalcAmp says (GeoAc_ConfigureCalcAmp(true)); CalcAmp
    // then only selects what is printed
    const bool amp_eqs = kRng ? true : CalcAmp;

    geoac_ctx* ctx = nullptr;
    int rc = geoac_create(&ctx, kEq, 0);
    if(rc){ cout << kName << ": " << geoac_strerror(rc) << '\n'; return 2; }
    if(apply_gpu_opts(ctx)) return 2;
    if(kRng) rc = geoac_upload_atmo_3d(ctx, grid.nx, grid.ny, grid.nz, grid.x.data(), grid.y.data(), grid.z.data(), grid.T.data(), grid.u.data(), grid.v.data(), grid.rho.data());
    else     rc = geoac_upload_atmo_1d(ctx, prof.n, prof.x.data(), prof.T.data(), prof.u.data(), prof.v.data(), prof.rho.data(), prof.sl.data());
    if(rc){ cout << kName << ": " << geoac_last_error(ctx) << '\n'; return 2; }
    P.z_grnd = z_grnd; P.tweak_abs = tweak_abs; P.freq = freq; P.calc_amp = amp_eqs ? 1 : 0;
    P.mode = GEOAC_MODE_WRITE_RAYS | GEOAC_MODE_INTERACTIVE | (WriteCaustics ? GEOAC_MODE_WRITE_CAUSTICS : 0);
    P.sample_stride = (kEq == GEOAC_EQ_3D || kRngS) ? 10 : 25;       // GeoAc3D_main.cpp:407, GeoAcGlobal.RngDep_main.cpp:462
    if(kSph){ P.src[0] = z_src; P.src[1] = src_a; P.src[2] = src_b; }
    else if(kEq == GEOAC_EQ_3D){ P.src[0] = 0.0; P.src[1] = 0.0; P.src[2] = z_src; }
    else if(kRngC){ P.src[0] = src_a; P.src[1] = src_b; P.src[2] = z_src; }
    else { P.src[0] = z_src; P.src[1] = 0.0; P.src[2] = 0.0; }

    double theta = 0.0, phi = 0.0;
    int bounces = 0;
    char keepgoing = 'y';
    ofstream raypath, caustics;                                       // one stream object each for the whole session: a setprecision sticks
    uint64_t steps_total = 0;
    long n_rays = 0;
    const double r_earth = 6370.0;

    while(keepgoing == 'y' || keepgoing == 'Y'){
        raypath.open("raypath.dat");
        if(kEq == GEOAC_EQ_GLOBAL)      raypath << "# z [km]" << '\t' << "lat [deg]" << '\t' << "lon [deg]";
        else if(kRngS)                  raypath << "# z [km]" << '\t' << "Lat [deg]" << '\t' << "Long [deg]";
        else if(kCart3)                 raypath << "# x [km]" << '\t' << "y [km]" << '\t' << "z [km]";
        else                            raypath << "# r [km]" << '\t' << "z [km]";
        raypath << '\t' << "Geo. Atten. [dB]" << '\t' << "Atmo. Atten. [dB]" << '\t' << "Travel Time [s]" << '\n';
        if(WriteCaustics){
            caustics.open("caustics.dat");
            if(kEq == GEOAC_EQ_GLOBAL)  caustics << "# z [km]" << '\t' << "lat [deg]" << '\t' << "lon [deg]";
            else if(kEq == GEOAC_EQ_2D) caustics << "# r [km]" << '\t' << "z [km]";
            else                        caustics << "# x [km]" << '\t' << "y [km]" << '\t' << "z [km]";     // also the spherical grid main (:432-435)
            caustics << '\t' << "Travel Time [s]" << '\n';
        }

        cout << '\t' << "Enter inclination angle [degrees]: ";  cin >> theta;
        cout << '\t' << "Enter azimuth angle [degrees]: ";      cin >> phi;
        cout << '\t' << "Enter number of bounces: ";            cin >> bounces;
        if(!cin){ raypath.close(); if(WriteCaustics) caustics.close(); break; }      // end of input: leave (the reference would spin on its last answer)
        cout << '\n';
        cout << '\t' << "Plotting ray path w/ theta = " << theta << ", phi = " << phi << '\n';

        if(bounces < 0) bounces = 0;
        P.bounces = bounces;
        rc = geoac_set_params(ctx, &P);
        if(rc){ cout << kName << ": " << geoac_last_error(ctx) << '\n'; geoac_destroy(ctx); return 2; }
        const int legs = bounces + 1;
        vector<double> rec((size_t)legs * GEOAC_REC_STRIDE, 0.0);
        uint64_t steps = 0;
        rc = geoac_fan_run(ctx, 1, &theta, &phi, rec.data(), &steps);
        if(rc){ cout << kName << ": " << geoac_last_error(ctx) << '\n'; geoac_destroy(ctx); return 2; }
        steps_total += steps; n_rays++;
        int64_t ns = 0;
        geoac_fan_sample_count(ctx, &ns);
        vector<double> smp((size_t)max<int64_t>(ns, 1) * GEOAC_SMP_STRIDE);
        if(ns > 0 && geoac_fan_fetch_samples(ctx, smp.data(), ns)){ cout << kName << ": " << geoac_last_error(ctx) << '\n'; geoac_destroy(ctx); return 2; }

        for(int64_t q = 0; q < ns; q++){
            const double* S = &smp[(size_t)q * GEOAC_SMP_STRIDE];
            const int kind = (int)S[GEOAC_SMP_KIND];
            const double* v = S + GEOAC_SMP_V0;
            if(kind == 0){
                const double geo = CalcAmp ? v[kCart3 || kSph ? 3 : 2] : 0.0;
                if(kSph){
                    raypath << v[0];
                    raypath << '\t' << setprecision(8) << v[1];
                    raypath << '\t' << setprecision(8) << v[2];
                    raypath << '\t' << geo << '\t' << v[4] << '\t' << v[5] << '\n';
                } else if(kCart3){
                    raypath << v[0] << '\t' << v[1] << '\t' << v[2] << '\t' << geo << '\t' << v[4] << '\t' << v[5] << '\n';
                } else {
                    raypath << v[0] << '\t' << v[1] << '\t' << geo << '\t' << v[3] << '\t' << v[4] << '\n';
                }
            } else if(WriteCaustics){
                if(kEq == GEOAC_EQ_GLOBAL){                          // latitude twice, blank line after each (GeoAcGlobal_main.cpp:451-457)
                    caustics << v[0];
                    caustics << '\t' << setprecision(8) << v[1];
                    caustics << '\t' << setprecision(8) << v[1];
                    caustics << '\t' << v[3] << '\n';
                    caustics << '\n';
                } else if(kRngS){                                    // raw radius and radians, max(lon, 0) (GeoAcGlobal.RngDep_main.cpp:472-478)
                    caustics << v[0] + r_earth;
                    caustics << '\t' << v[1] * Pi / 180.0;
                    caustics << '\t' << max(v[2] * Pi / 180.0, 0.0);
                    caustics << '\t' << v[3] << '\n';
                    caustics << '\n';
                } else if(kRngC){
                    caustics << v[0] << '\t' << v[1] << '\t' << max(v[2], 0.0) << '\t' << v[4] << '\n';
                } else if(kEq == GEOAC_EQ_3D){
                    caustics << v[0] << '\t' << v[1] << '\t' << v[2] << '\t' << v[3] << '\n';
                } else {
                    caustics << v[0] << '\t' << v[1] << '\t' << v[2] << '\n';
                }
            }
        }

        // arrival summary from the record of the last leg (solution[k] of the last propagation)
        bool broke = false;
        double z_max = 0.0;
        for(int b = 0; b < legs; b++){
            const double* R = &rec[(size_t)b * GEOAC_REC_STRIDE];
            if(R[GEOAC_REC_VALID] == 0.0){ broke = true; break; }
            z_max = max(z_max, R[GEOAC_REC_TURN]);
        }
        const double* R = &rec[(size_t)(legs - 1) * GEOAC_REC_STRIDE];
        const double* yk = R + GEOAC_REC_STATE;
        if(broke){
            if(kRngS) cout << '\t' << '\t' << "Ray path does not return to the ground " << '\n' << '\n';
            else      cout << '\t' << '\t' << "Ray path does not return to the ground." << '\n' << '\n';
        } else if(kSph){
            const double lat0 = src_a * Pi / 180.0, lon0 = src_b * Pi / 180.0;
            const double arg1 = sin(yk[2] - lon0);
            const double arg2 = cos(lat0) * tan(yk[1]) - sin(lat0) * cos(yk[2] - lon0);
            const double bearing = atan2(arg1, arg2) * 180.0 / Pi;
            cout << '\t' << '\t' << "Ray path arrived at " << yk[1] * 180.0 / Pi << " degrees latitude, " << yk[2] * 180.0 / Pi << " degrees longitude." << '\n';
            cout << '\t' << '\t' << "Arrival range = " << R[GEOAC_REC_RANGE] << " km at azimuth " << bearing << " degrees from N." << '\n';
            if(CalcAmp) cout << '\t' << '\t' << "Geometric attenuation = " << 20.0 * log10(R[GEOAC_REC_AMP]) << " dB." << '\n';
            cout << '\t' << '\t' << "Atmospheric attenuation = " << -R[GEOAC_REC_ATTEN] << " dB." << '\n';
            cout << '\t' << '\t' << "Arrival celerity = " << R[GEOAC_REC_RANGE] / R[GEOAC_REC_TTIME] << "km/sec." << '\n';
            cout << '\t' << '\t' << "Back azimuth of the arrival = " << R[GEOAC_REC_BACKAZ] << " degrees from N. " << '\n' << '\n';
        } else if(kCart3){
            const double range = sqrt(yk[0] * yk[0] + yk[1] * yk[1]);
            double back_az;
            if(kRngC) back_az = R[GEOAC_REC_BACKAZ];
            else { back_az = phi - 180.0; while(back_az < -180.0) back_az += 360.0; while(back_az > 180.0) back_az -= 360.0; }
            cout << '\t' << '\t' << "Ray path arrived at " << yk[0] << " km E-W, " << yk[1] << " km N-S." << '\n';
            cout << '\t' << '\t' << "Arrival range = " << range << " km at azimuth " << atan2(yk[1], yk[0]) * 180.0 / Pi << " degrees from N." << '\n';
            if(CalcAmp) cout << '\t' << '\t' << "Geometric attenuation = " << 20.0 * log10(R[GEOAC_REC_AMP]) << " dB." << '\n';
            cout << '\t' << '\t' << "Atmospheric attenuation = " << -R[GEOAC_REC_ATTEN] << " dB." << '\n';
            cout << '\t' << '\t' << "Arrival celerity = " << range / R[GEOAC_REC_TTIME] << " km/sec." << '\n';
            cout << '\t' << '\t' << "Turning height of the ray = " << z_max << " km." << '\n';
            cout << '\t' << '\t' << "Back azimuth of the arrival = " << back_az << " degrees (relative to N). " << '\n' << '\n';
        } else {
            cout << '\t' << '\t' << "Arrival Range = " << yk[0] << '\n';
            if(CalcAmp) cout << '\t' << '\t' << "Geometric Attenuation = " << 20.0 * log10(R[GEOAC_REC_AMP]) << " dB." << '\n';
            cout << '\t' << '\t' << "Atmospheric Attenuation = " << -R[GEOAC_REC_ATTEN] << " dB." << '\n';
            cout << '\t' << '\t' << "Turning Height = " << z_max << "km." << '\n';
            cout << '\t' << '\t' << "Travel Time = " << R[GEOAC_REC_TTIME] << "sec." << '\n';
            cout << '\t' << '\t' << "Arrival Celerity = " << yk[0] / R[GEOAC_REC_TTIME] << "km/sec." << '\n' << '\n';
        }

        raypath.close();
        if(WriteCaustics) caustics.close();

        cout << "Continue plotting other ray paths? (y/n): ";
        if(!(cin >> keepgoing)) break;
    }
    geoac_destroy(ctx);
    cerr << kName << ": " << n_rays << " rays, " << steps_total << " RK4 ray-steps on the GPU" << '\n';
    return 0;
}

// ---- -eig_search / -eig_direct (GeoAcGlobal_main.cpp:496-660, GeoAcGlobal.RngDep_main.cpp:518-693, GeoAc3D_main.cpp:458-610,
//      GeoAc3D.RngDep_main.cpp:497-670); the searches themselves are geoac_eig_search / geoac_eig_direct ----
static int run_eig(char* inputs[], int count, bool direct){
    double Source_Loc[3] = {30.0, 0.0, 0.0};
    double Receiver_Loc[2] = {30.0, -2.5};
    if(kCart3){ Source_Loc[0] = 0.0; Source_Loc[1] = 0.0; Receiver_Loc[0] = -250.0; Receiver_Loc[1] = 0.0; }
    double theta_est = kCart3 ? 0.5 : 10.0, phi_est = 45.0;
    int bounces = 0;
    bool verbose_output = false;
    const char* ProfileFormat = "zTuvdp";
    double z_grnd = 0.0, tweak_abs = 0.3, freq = 0.1;
    char input_check;
    const int arg0 = kRng ? 5 : 3;
    geoac_eig_params E; geoac_eig_default_params(&E);
    geoac_params P; geoac_default_params(kEq, &P);

    for(int i = arg0; i < count; i++){
        if(strncmp(inputs[i], "profile_format=", 15) == 0) ProfileFormat = inputs[i] + 15;
        else if(kRng && strncmp(inputs[i], "z_grnd=", 7) == 0) z_grnd = atof(inputs[i] + 7);        // the grid mains know z_grnd when they load
    }
    Profile prof; Grid grid;
    if(kRng){
        if(geoac_grid_dims(inputs[2], inputs[3], inputs[4], &grid.nx, &grid.ny, &grid.nz)){ cout << "Error opening file, check file name" << '\n'; return 1; }
        const size_t nn = (size_t)grid.nx * grid.ny * grid.nz;
        grid.x.resize(grid.nx); grid.y.resize(grid.ny); grid.z.resize(grid.nz);
        grid.T.resize(nn); grid.u.resize(nn); grid.v.resize(nn); grid.rho.resize(nn);
        if(geoac_grid_load_eq(kEq, inputs[2], inputs[3], inputs[4], ProfileFormat, z_grnd, grid.nx, grid.ny, grid.nz, grid.x.data(), grid.y.data(), grid.z.data(),
                              grid.T.data(), grid.u.data(), grid.v.data(), grid.rho.data())){ cout << "Error opening file, check file name" << '\n'; return 1; }
        if(kRngS){
            const double t0 = grid.x[0], t1 = grid.x[(size_t)grid.nx - 1], p0 = grid.y[0], p1 = grid.y[(size_t)grid.ny - 1];
            Source_Loc[0] = (t0 + t1) / 2.0 * 180.0 / Pi; Source_Loc[1] = (p0 + p1) / 2.0 * 180.0 / Pi; Source_Loc[2] = 0.0;     // :537-539
            Receiver_Loc[0] = (t0 + t1) * 0.75 * 180.0 / Pi; Receiver_Loc[1] = (p0 + p1) * 0.75 * 180.0 / Pi;               // :541-542
        }
    } else {
        prof.n = geoac_met_rows(inputs[2]);
        if(prof.n < 3){ cout << "Error opening file, check file name" << '\n'; return 1; }
        prof.x.resize(prof.n); prof.T.resize(prof.n); prof.u.resize(prof.n); prof.v.resize(prof.n); prof.rho.resize(prof.n);
        prof.sl.resize(4 * (size_t)prof.n);
        if(geoac_met_load(inputs[2], ProfileFormat, kEq, prof.n, prof.x.data(), prof.T.data(), prof.u.data(), prof.v.data(), prof.rho.data()) != prof.n){
            cout << "Unrecognized profile option: " << ProfileFormat << ".  Valid options are: zTuvdp and zuvwTdp" << '\n'; return 1; }
        geoac_natural_spline_slopes(prof.n, prof.x.data(), prof.T.data(),   &prof.sl[0]);
        geoac_natural_spline_slopes(prof.n, prof.x.data(), prof.u.data(),   &prof.sl[prof.n]);
        geoac_natural_spline_slopes(prof.n, prof.x.data(), prof.v.data(),   &prof.sl[2 * (size_t)prof.n]);
        geoac_natural_spline_slopes(prof.n, prof.x.data(), prof.rho.data(), &prof.sl[3 * (size_t)prof.n]);
        P.vert_limit = prof.x[prof.n - 1];
    }
    for(int i = arg0; i < count; i++){
        const char* a = inputs[i];
        if(!direct && strncmp(a, "theta_min=", 10) == 0){ E.theta_min = atof(a + 10); }
        else if(!direct && strncmp(a, "theta_max=", 10) == 0){ E.theta_max = atof(a + 10); }
        else if(!direct && strncmp(a, "bnc_min=", 8) == 0){ E.bnc_min = atoi(a + 8); }
        else if(!direct && strncmp(a, "bnc_max=", 8) == 0){ E.bnc_max = atoi(a + 8); }
        else if(strncmp(a, "bounces=", 8) == 0){ E.bnc_min = atoi(a + 8); E.bnc_max = atoi(a + 8); bounces = atoi(a + 8); }
        else if(direct && strncmp(a, "theta_est=", 10) == 0){ theta_est = atof(a + 10); }
        else if(direct && strncmp(a, "phi_est=", 8) == 0){ }                                           // read after the bearing is known
        else if(kSph && strncmp(a, "lat_src=", 8) == 0){ Source_Loc[0] = atof(a + 8); }
        else if(kSph && strncmp(a, "lon_src=", 8) == 0){ Source_Loc[1] = atof(a + 8); }
        else if(kRngC && strncmp(a, "x_src=", 6) == 0){ Source_Loc[0] = atof(a + 6); }
        else if(kRngC && strncmp(a, "y_src=", 6) == 0){ Source_Loc[1] = atof(a + 6); }
        else if(strncmp(a, "z_src=", 6) == 0){ Source_Loc[2] = atof(a + 6); }
        else if(kSph && strncmp(a, "lat_rcvr=", 9) == 0){ Receiver_Loc[0] = atof(a + 9); }
        else if(kSph && strncmp(a, "lon_rcvr=", 9) == 0){ Receiver_Loc[1] = atof(a + 9); }
        else if(kCart3 && strncmp(a, "x_rcvr=", 7) == 0){ Receiver_Loc[0] = atof(a + 7); }
        else if(kCart3 && strncmp(a, "y_rcvr=", 7) == 0){ Receiver_Loc[1] = atof(a + 7); }
        else if(strncmp(a, "Verbose=", 8) == 0 || strncmp(a, "verbose=", 8) == 0){ verbose_output = string2bool(a + 8); }
        else if(!direct && strncmp(a, "azimuth_err_lim=", 16) == 0){ E.azimuth_err_lim = atof(a + 16); }
        else if(strncmp(a, "iterations=", 11) == 0){ E.iterations = (int)atof(a + 11); }
        else if(strncmp(a, "freq=", 5) == 0){ freq = atof(a + 5); }
        else if(strncmp(a, "abs_coeff=", 10) == 0){ tweak_abs = max(0.0, atof(a + 10)); }
        else if(strncmp(a, "z_grnd=", 7) == 0){ z_grnd = atof(a + 7); }
        else if(strncmp(a, "profile_format=", 15) == 0){ }
        else if(kRngC && strncmp(a, "alt_max=", 8) == 0){ }                                           // overwritten by GeoAc_SetPropRegion after the load (Q9)
        else if(kRngC && (strncmp(a, "x_min=", 6) == 0 || strncmp(a, "x_max=", 6) == 0 || strncmp(a, "y_min=", 6) == 0 || strncmp(a, "y_max=", 6) == 0)){ }
        else if(strncmp(a, "alt_max=", 8) == 0){ P.vert_limit = atof(a + 8); }
        else if(!kRng && strncmp(a, "rng_max=", 8) == 0){ P.range_limit = atof(a + 8); }
        else if(kRngS && strncmp(a, "lat_min=", 8) == 0){ P.xy_limits[0] = atof(a + 8); }
        else if(kRngS && strncmp(a, "lat_max=", 8) == 0){ P.xy_limits[1] = atof(a + 8); }
        else if(kRngS && strncmp(a, "lon_min=", 8) == 0){ P.xy_limits[2] = atof(a + 8); }
        else if(kRngS && strncmp(a, "lon_max=", 8) == 0){ P.xy_limits[3] = atof(a + 8); }
        else if(strncmp(a, "gpu_", 4) == 0){ }                        // arguments of this GPU build (parse_gpu_args)
        else {
            cout << "***WARNING*** Unrecognized parameter entry: " << a << '\n';
            cout << "Continue? (y/n):"; cin >> input_check;
            if(input_check != 'y' && input_check != 'Y') return 0;
        }
    }
    Source_Loc[2] = max(Source_Loc[2], z_grnd);
    E.verbose = verbose_output ? 1 : 0;

    geoac_ctx* ctx = nullptr;
    int rc = geoac_create(&ctx, kEq, 0);
    if(rc){ cout << kName << ": " << geoac_strerror(rc) << '\n'; return 2; }
    if(apply_gpu_opts(ctx)) return 2;
    if(kRng) rc = geoac_upload_atmo_3d(ctx, grid.nx, grid.ny, grid.nz, grid.x.data(), grid.y.data(), grid.z.data(), grid.T.data(), grid.u.data(), grid.v.data(), grid.rho.data());
    else     rc = geoac_upload_atmo_1d(ctx, prof.n, prof.x.data(), prof.T.data(), prof.u.data(), prof.v.data(), prof.rho.data(), prof.sl.data());
    if(rc){ cout << kName << ": " << geoac_last_error(ctx) << '\n'; return 2; }
    P.z_grnd = z_grnd; P.tweak_abs = tweak_abs; P.freq = freq;
    if(kSph){ P.src[0] = Source_Loc[2]; P.src[1] = Source_Loc[0]; P.src[2] = Source_Loc[1]; }
    else    { P.src[0] = Source_Loc[0]; P.src[1] = Source_Loc[1]; P.src[2] = Source_Loc[2]; }
    rc = geoac_set_params(ctx, &P);
    if(rc){ cout << kName << ": " << geoac_last_error(ctx) << '\n'; return 2; }

    char file_title[64];
    { int m = 0; for(; m < 50 && inputs[2][m] != '\0' && inputs[2][m] != '.'; m++) file_title[m] = inputs[2][m]; file_title[m] = '\0'; }
    char output_buffer[96];

    geoac_eig_result* res = nullptr;
    ofstream results;
    const auto t_eig0 = std::chrono::steady_clock::now();
    if(direct){
        if(kSph){       // bearing to the receiver unless phi_est= was given (GeoAcGlobal_main.cpp:636-643)
            double term1 = sin((Receiver_Loc[1] - Source_Loc[1]) * Pi / 180.0);
            double term2 = cos(Source_Loc[0] * Pi / 180.0) * tan(Receiver_Loc[0] * Pi / 180.0) - sin(Source_Loc[0] * Pi / 180.0) * cos((Receiver_Loc[1] - Source_Loc[1]) * Pi / 180.0);
            phi_est = 90.0 - atan2(term1, term2) * 180.0 / Pi;
        } else {        // GeoAc3D_main.cpp:588: azimuth of the receiver position itself
            phi_est = 180.0 / 3.14159 * atan2(Receiver_Loc[1], Receiver_Loc[0]);
        }
        for(int i = arg0; i < count; i++) if(strncmp(inputs[i], "phi_est=", 8) == 0) phi_est = 90.0 - atof(inputs[i] + 8);
        double phi_from_north = 90.0 - phi_est;
        rc = geoac_eig_direct(ctx, &E, 1, Receiver_Loc, &theta_est, &phi_from_north, bounces, &res);
    } else {
        sprintf(output_buffer, "%s_results.dat", file_title);
        results.open(output_buffer);
        results << kName << " - Eigenray Run Summary:" << '\n';
        results << '\t' << "Profile used: " << inputs[2] << '\n';
        results << '\t' << (kSph ? "Source Location (lat, lon, elev) : (" : "Source Location (kilometers) : (") << Source_Loc[0] << ", " << Source_Loc[1] << ", " << Source_Loc[2] << ")." << '\n';
        results << '\t' << (kSph ? "Receiver Location (lat, lon, elev) : (" : "Receiver Location (kilometers) : (") << Receiver_Loc[0] << ", " << Receiver_Loc[1] << ", " << z_grnd << ")." << '\n';
        results << '\t' << "Inclination range (degrees): " << E.theta_min << " - " << E.theta_max << "." << '\n';
        results << '\t' << "Ground reflection (bounce) limits: " << E.bnc_min << " - " << E.bnc_max << "." << '\n' << '\n';
        rc = geoac_eig_search(ctx, &E, 1, Receiver_Loc, &res);
    }
    if(rc){ cout << kName << ": " << geoac_last_error(ctx) << '\n'; geoac_destroy(ctx); return 2; }
    cout << geoac_eig_log(res, 0);

    const int64_t ne = geoac_eig_count(res), ns = geoac_eig_sample_count(res);
    vector<double> eig((size_t)max<int64_t>(ne, 1) * GEOAC_EIG_STRIDE), smp((size_t)max<int64_t>(ns, 1) * GEOAC_SMP_STRIDE);
    if(ne) geoac_eig_fetch(res, eig.data());
    if(ns) geoac_eig_fetch_samples(res, smp.data());
    for(int64_t e = 0; e < ne; e++){
        const double* V = &eig[(size_t)e * GEOAC_EIG_STRIDE];
        sprintf(output_buffer, "%s_Eigenray-%i.dat", file_title, (int)V[GEOAC_EIG_INDEX]);
        ofstream raypath; raypath.open(output_buffer);
        if(kSph) raypath << "# z [km]" << '\t' << "lat [deg]" << '\t' << "lon [deg]";
        else     raypath << "# x [km]" << '\t' << "y [km]" << '\t' << "z [km]";
        raypath << '\t' << "Geo. Atten. [dB]" << '\t' << "Atmo. Atten. [dB]" << '\t' << "Travel Time [s]" << '\n';
        const size_t s0 = (size_t)V[GEOAC_EIG_SMP0], sn = (size_t)V[GEOAC_EIG_NSMP];
        for(size_t q = s0; q < s0 + sn; q++){
            const double* S = &smp[q * GEOAC_SMP_STRIDE];
            const double* v = S + GEOAC_SMP_V0;
            const bool first_leg = ((int)S[GEOAC_SMP_LEG] == 0);
            if(kSph && first_leg){
                // Eigenray.Global.cpp:207-214: lat/lon with 8 digits through Pi ...
                raypath << v[0];
                raypath << '\t' << setprecision(8) << v[1];
                raypath << '\t' << setprecision(8) << v[2];
                raypath << '\t' << v[3] << '\t' << v[4] << '\t' << v[5] << '\n';
            } else if(kSph){
                // ... :226-233: later legs through 3.14159 and the attenuation without its minus sign
                raypath << v[0];
                raypath << '\t' << (v[1] * Pi / 180.0) * 180.0 / 3.14159;
                raypath << '\t' << (v[2] * Pi / 180.0) * 180.0 / 3.14159;
                raypath << '\t' << v[3] << '\t' << -v[4] << '\t' << v[5] << '\n';
            } else {
                // Eigenray.cpp:222-246: x, y, z; later legs print the attenuation without its minus sign
                raypath << v[0] << '\t' << v[1] << '\t' << v[2] << '\t' << v[3] << '\t' << (first_leg ? v[4] : -v[4]) << '\t' << v[5] << '\n';
            }
        }
        raypath.close();
        if(!direct){
            results << "Eigenray-" << (int)V[GEOAC_EIG_INDEX] << ".  " << (int)V[GEOAC_EIG_BOUNCES] << " bounce(s)." << '\n';
            results << '\t' << "theta, phi = " << setprecision(8) << V[GEOAC_EIG_THETA] << ", " << V[GEOAC_EIG_PHI] << " degrees." << '\n';
            results << '\t' << "Travel Time = " << V[GEOAC_EIG_TTIME] << " seconds." << '\n';
            results << '\t' << "Celerity = " << V[GEOAC_EIG_CELERITY] << " km/s." << '\n';
            results << '\t' << "Amplitude (geometric) = " << V[GEOAC_EIG_AMP_DB] << " dB." << '\n';
            results << '\t' << "Atmospheric attenuation = " << V[GEOAC_EIG_ATTEN_DB] << " dB." << '\n';
            results << '\t' << "Arrival inclination = " << V[GEOAC_EIG_INCL] << " degrees." << '\n';
            if(kSph){
                results << '\t' << "Bearing to source = " << V[GEOAC_EIG_BEARING] << " degrees." << '\n';
                results << '\t' << "Back azimuth of arrival = " << V[GEOAC_EIG_BACKAZ] << " degrees." << '\n';
            } else {
                results << '\t' << "Azimuth to source = " << V[GEOAC_EIG_BEARING] << '\n';
                results << '\t' << "Back azimuth of arrival = " << V[GEOAC_EIG_BACKAZ] << '\n';
            }
            results << '\t' << "Azimuth deviation = " << V[GEOAC_EIG_AZDEV] << " degrees." << '\n' << '\n';
        }
    }
    if(!direct) results.close();
    uint64_t st[8]; geoac_eig_stats_ex(res, st);
    cerr << kName << ": " << st[1] << " rays in " << st[0] << " fan launches (" << st[3] << " rounds), " << st[2] << " RK4 ray-steps on the GPU, " << st[4]
         << " of them along the critical path (the longest ray of every launch)" << '\n';
    write_stats(direct ? "-eig_direct" : "-eig_search", (long)st[1], st[2], std::chrono::duration<double>(std::chrono::steady_clock::now() - t_eig0).count(),
                vector<int>(1, 0), vector<uint64_t>(1, st[1]), vector<uint64_t>(1, st[2]), vector<uint64_t>(1, st[0]));
    geoac_eig_free(res);
    geoac_destroy(ctx);
    return 0;
}

int main(int argc, char* argv[]){
    if(argc >= 2 && strcmp(argv[1], "-format_selftest") == 0) return format_selftest(argc >= 3 ? atol(argv[2]) : 1000000L);
    if(argc < (kRng ? 5 : 3)){ usage(); return 0; }
    parse_gpu_args(argc, argv);
    if(strncmp(argv[1], "-prop", 5) == 0) return run_prop(argv, argc);
    if(kEq != GEOAC_EQ_2D && strncmp(argv[1], "-eig_search", 11) == 0) return run_eig(argv, argc, false);
    if(kEq != GEOAC_EQ_2D && strncmp(argv[1], "-eig_direct", 11) == 0) return run_eig(argv, argc, true);
    if(strncmp(argv[1], "-interactive", 12) == 0) return run_interactive(argv, argc);
    if(strncmp(argv[1], "-eig_search", 11) == 0 || strncmp(argv[1], "-eig_direct", 11) == 0){
        cout << kName << ": option " << argv[1] << " is not part of this GPU build." << '\n';
        return 3;
    }
    cout << "Unrecognized option." << '\n';
    return 0;
}
