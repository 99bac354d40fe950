"""step latency of the two-lane RK4 kernel against the number of waves on the chip: fans of one inclination (0.5 deg, the longest rays),
n azimuths; prints the pass time divided by the steps of the longest ray.  usage: perf_pair_occupancy.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geoac_amd as G
import harness as H
for n_az in (32, 360, 3240, 8192, 16200):
    ctx = G.FanContext(G.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO)
    ctx.set_params(bounces=2, calc_amp=1, mode=0, src=(0.0, 30.0, 0.0))
    ph = -180.0 + 360.0 * np.arange(n_az) / n_az
    th = np.full(n_az, 0.5)
    ctx.set_angles(th, ph); ctx.launch()
    t0 = time.perf_counter(); ctx.launch(); dt = time.perf_counter() - t0
    rec, steps = ctx.fetch()
    longest = rec[:, :, 1].sum(axis=1).max()
    tm = ctx.timing()
    print(f"{n_az:6d} rays ({n_az * 2 // 64 + 1:4d} waves): {dt * 1e3:7.1f} ms, longest ray {int(longest)} steps -> {dt / longest * 1e6:.3f} us per step (rk4 {tm['ms_rk4']:.1f} ms, epochs {tm['epochs']})")
    ctx.close()
