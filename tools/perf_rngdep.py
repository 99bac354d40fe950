"""steps/s of the GPU range-dependent paths on the synthetic 5x5 grids (tests/rngdep_data.py);
usage: perf_rngdep.py [n_az] [n_incl] [3d|global] [thin: 4 = 350 levels (default), 1 = 1400]"""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geoac_amd as G
import rngdep_data as RD

n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n_in = int(sys.argv[2]) if len(sys.argv) > 2 else 40
which = sys.argv[3] if len(sys.argv) > 3 else "3d"
thin = int(sys.argv[4]) if len(sys.argv) > 4 else 4
if which == "global":
    grid = RD.write_grid_global(os.path.join(tempfile.gettempdir(), "ggp"), short_paths=False)
    eq, src = G.EQ_GLOBAL_RNGDEP, (0.0, 31.0, 0.0)
else:
    grid = RD.write_grid(os.path.join(tempfile.gettempdir(), f"gdp{thin}"), short_paths=False, thin=thin)
    eq, src = G.EQ_3D_RNGDEP, (0.0, 0.0, 0.0)
th, ph = G.fan_enumerate(theta_min=1.0, theta_max=1.0 + (n_in - 1) * 1.0, theta_step=1.0, phi_min=-180.0, phi_max=-180.0 + (n_az - 1) * (360.0 / n_az), phi_step=360.0 / n_az)
for amp in (1, 0):
    ctx = G.FanContext(eq, device=0)
    ctx.load_grid(*grid)
    ctx.set_params(bounces=1, calc_amp=amp, mode=0, src=src)
    ctx.set_angles(th, ph)
    ctx.launch()
    t0 = time.perf_counter(); ctx.launch(); dt = time.perf_counter() - t0
    tm = ctx.timing()
    print(f"{which} CalcAmp={amp}: {len(th)} rays, {ctx.total_steps()} steps, {dt*1e3:.1f} ms, {ctx.total_steps()/dt:.3e} steps/s  (rk4 {tm['ms_rk4']:.1f} ms, post {tm['ms_post']:.1f} ms, epochs {tm['epochs']})")
    ctx.close()
