/* CPU-only sanitizer driver (tests/test_sanitizers.py): the plain-C oracle (test infrastructure) under -fsanitize=address,undefined:
 * a few rays of every 1-D equation set in every output mode, the atmosphere and absorption probes. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../oracle/geoac_oracle.h"

int main(int argc, char** argv){
    if(argc < 2) return 2;
    for(int eq = 0; eq <= 2; eq++){
        orc_ctx* c = orc_create(eq);
        if(!c || orc_load(c, argv[1], "zTuvdp") != 1400){ fprintf(stderr, "load failed\n"); return 1; }
        double th[3] = { 5.0, 22.0, 41.0 }, ph[3] = { -90.0, 10.0, 135.0 };
        for(int mode = 0; mode <= 3; mode += (mode == 1 ? 2 : 1)){
            ref_fan_cfg cfg; memset(&cfg, 0, sizeof cfg);
            cfg.tweak_abs = 0.3; cfg.freq = 0.1; cfg.vert_limit = NAN; cfg.range_limit = NAN;
            cfg.src[0] = 0.0; cfg.src[1] = (eq == 2) ? 30.0 : 0.0; cfg.src[2] = 0.0;
            cfg.bounces = 1; cfg.calc_amp = 1; cfg.mode = mode;
            for(int q = 0; q < 4; q++) cfg.xy_limits[q] = NAN;
            double* rec = (double*)calloc((size_t)3 * 2 * 32, sizeof(double));
            double* smp = (double*)calloc((size_t)20000 * 10, sizeof(double));
            int64_t ns = 0;
            int64_t steps = orc_fan(c, &cfg, 3, th, ph, rec, smp, 20000, &ns);
            if(steps <= 0 || !isfinite(rec[3])){ fprintf(stderr, "fan failed eq %d mode %d\n", eq, mode); return 1; }
            free(rec); free(smp);
        }
        double x[4] = { 0.0, 0.3, 70.0, 139.0 }, o9[36], rho[4], f[4] = { 0.01, 0.1, 1.0, 10.0 }, al[4];
        if(eq == 2) for(int i = 0; i < 4; i++) x[i] += 6370.0;
        orc_atmo_probe(c, 4, x, o9, rho);
        orc_absorption_probe(c, 4, x, f, 0.0, 0.3, al);
        for(int i = 0; i < 4; i++) if(!isfinite(al[i]) || !isfinite(o9[9 * i])){ fprintf(stderr, "probe failed\n"); return 1; }
        orc_destroy(c);
    }
    printf("san_oracle_driver ok\n");
    return 0;
}
