"""step latency of the RK4 kernels of the stratified Global set with amplitudes: a fan of the longest rays (inclination 0.5 deg, n azimuths);
prints pass time / steps of the longest ray for the launch plans given as GEOAC_DUO values (0 = the one-wave kernels).
usage: perf_duo.py [n_az] [duo values ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geoac_amd as G
import harness as H
n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 3240
for v in (sys.argv[2:] or ["0", "1", "32", "66"]):
    ctx = G.FanContext(G.EQ_GLOBAL, device=0, options={"DUO": v}); ctx.load_met(H.TOYATMO)
    ctx.set_params(bounces=2, calc_amp=1, mode=0, src=(0.0, 30.0, 0.0))
    ph = -180.0 + 360.0 * np.arange(n_az) / n_az
    th = np.full(n_az, 0.5)
    ctx.set_angles(th, ph); ctx.launch()
    t0 = time.perf_counter(); ctx.launch(); dt = time.perf_counter() - t0
    rec, steps = ctx.fetch()
    longest = rec[:, :, 1].sum(axis=1).max()
    tm = ctx.timing()
    print(f"GEOAC_DUO={v:>3s} {n_az:6d} rays: {dt * 1e3:7.1f} ms, longest ray {int(longest)} steps -> {dt / longest * 1e6:.3f} us per step (rk4 {tm['ms_rk4']:.1f} ms, epochs {tm['epochs']})", flush=True)
    ctx.close()
