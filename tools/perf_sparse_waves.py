"""does a two-lane RK4 wave with FEWER rays step faster?  Fans of one inclination, n azimuths, n <= 32 (one wave, part filled) and 64 (two waves): the pass time
divided by the steps of the longest ray.  (sparse critical waves.)  usage: perf_sparse_waves.py [theta ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geoac_amd as G
import harness as H
for theta in ([float(a) for a in sys.argv[1:]] or [0.5, 2.0, 4.0]):
    for n_az in (2, 8, 16, 32, 64):
        ctx = G.FanContext(G.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO)
        ctx.set_params(bounces=2, calc_amp=1, mode=0, src=(0.0, 30.0, 0.0))
        ph = -180.0 + 360.0 * np.arange(n_az) / n_az
        th = np.full(n_az, theta)
        ctx.set_angles(th, ph); ctx.launch()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); ctx.launch(); ts.append(time.perf_counter() - t0)
        rec, steps = ctx.fetch()
        longest = rec[:, :, 1].sum(axis=1).max()
        tm = ctx.timing()
        print(f"theta {theta:4.1f}  {n_az:3d} rays: {min(ts) * 1e3:7.2f} ms, longest ray {int(longest)} steps -> {min(ts) / longest * 1e6:.3f} us per step (rk4 {tm['ms_rk4']:.1f} ms, epochs {tm['epochs']})", flush=True)
        ctx.close()
