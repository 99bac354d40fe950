// CPU-only sanitizer driver (tests/test_sanitizers.py): the host-side set-up helpers of libgeoac_hip (geoac_amd/csrc/geoac_host.cpp:
// .met reader, natural-spline slopes, launch-angle enumeration, grid loader and grid table) compiled with -fsanitize=address,undefined
// and run over the fixtures, plus the error paths (missing files, bad format string, empty fan).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../include/geoac_hip.h"
#include "../../include/geoac_host.h"

#define CHECK(c) do { if(!(c)){ fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while(0)

int main(int argc, char** argv){
    if(argc < 5){ fprintf(stderr, "usage: san_host_driver ToyAtmo.met grid_prefix loc_x loc_y\n"); return 2; }
    const char* met = argv[1];
    // ---- 1-D profile ----
    for(int eq : {GEOAC_EQ_2D, GEOAC_EQ_3D, GEOAC_EQ_GLOBAL}){
        int n = geoac_met_rows(met);
        CHECK(n == 1400);
        std::vector<double> x(n), T(n), u(n), v(n), rho(n), sl(4 * (size_t)n);
        CHECK(geoac_met_load(met, "zTuvdp", eq, n, x.data(), T.data(), u.data(), v.data(), rho.data()) == n);
        CHECK(geoac_met_load_zg(met, "zTuvdp", eq, 0.3, n, x.data(), T.data(), u.data(), v.data(), rho.data()) == n);
        CHECK(geoac_met_load(met, "nonsense", eq, n, x.data(), T.data(), u.data(), v.data(), rho.data()) != n);
        CHECK(geoac_met_load(met, "zTuvdp", eq, 10, x.data(), T.data(), u.data(), v.data(), rho.data()) != n);     // capacity too small
        for(int f = 0; f < 4; f++) geoac_natural_spline_slopes(n, x.data(), (f == 0 ? T : f == 1 ? u : f == 2 ? v : rho).data(), &sl[(size_t)f * n]);
        for(double s : sl) CHECK(std::isfinite(s));
        double c[4];
        geoac_spline_segment_cubic(x[3], x[4], T[3], T[4], sl[3], sl[4], c, 1);
        CHECK(std::isfinite(c[0] + c[1] + c[2] + c[3]));
    }
    CHECK(geoac_met_rows("/nonexistent/file.met") <= 0);
    // ---- launch-angle enumeration (repeated addition, as the reference's loops) ----
    long n = geoac_fan_enumerate(0.5, 45.0, 0.5, -180.0, 179.0, 1.0, 0, nullptr, nullptr);
    CHECK(n == 32400);
    std::vector<double> th((size_t)n), ph((size_t)n);
    CHECK(geoac_fan_enumerate(0.5, 45.0, 0.5, -180.0, 179.0, 1.0, n, th.data(), ph.data()) == n);
    CHECK(th[0] == 0.5 && ph[(size_t)n - 1] == 179.0);
    CHECK(geoac_fan_enumerate(10.0, 5.0, 1.0, 0.0, 0.0, 1.0, 0, nullptr, nullptr) == 0);        // empty fan
    CHECK(geoac_fan_enumerate(0.05, 50.0, 0.05, -180.0, -180.0, 1.0, 0, nullptr, nullptr) >= 999);
    // ---- grid of profiles, both sets ----
    for(int eq : {GEOAC_EQ_3D_RNGDEP, GEOAC_EQ_GLOBAL_RNGDEP}){
        int nx = 0, ny = 0, nz = 0;
        CHECK(geoac_grid_dims(argv[2], argv[3], argv[4], &nx, &ny, &nz) == 0);
        CHECK(nx == 5 && ny == 5 && nz > 10);
        const size_t nn = (size_t)nx * ny * nz;
        std::vector<double> x(nx), y(ny), z(nz), T(nn), u(nn), v(nn), rho(nn);
        CHECK(geoac_grid_load_eq(eq, argv[2], argv[3], argv[4], "zTuvdp", 0.0, nx, ny, nz, x.data(), y.data(), z.data(), T.data(), u.data(), v.data(), rho.data()) == 0);
        CHECK(geoac_grid_load_eq(eq, argv[2], argv[3], argv[4], "bad", 0.0, nx, ny, nz, x.data(), y.data(), z.data(), T.data(), u.data(), v.data(), rho.data()) != 0);
        std::vector<double> tab(geoac_grid_table_size(nx, ny, nz));
        CHECK(geoac_grid_table_eq(eq, nx, ny, nz, x.data(), y.data(), z.data(), T.data(), u.data(), v.data(), rho.data(), tab.data()) == 0);
        for(int q = 0; q < 50; q++){
            double px = x[0] + (x[nx - 1] - x[0]) * (q / 49.0), py = y[0] + (y[ny - 1] - y[0]) * ((q * 7 % 50) / 49.0), pz = z[0] + (z[nz - 1] - z[0]) * ((q * 13 % 50) / 49.0);
            for(int f = 0; f < 4; f++) CHECK(std::isfinite(geoac_grid_eval_eq(eq, nx, ny, nz, x.data(), y.data(), z.data(), tab.data(), f, px, py, pz)));
        }
    }
    int a, b, c;
    CHECK(geoac_grid_dims("/nonexistent/p", argv[3], argv[4], &a, &b, &c) != 0);
    printf("san_host_driver ok\n");
    return 0;
}
