"""step time of one wave of shallow rays under the hand-off variants of k_rk4_trio (TRIO = 1 shipped; 3: the base wave never waits; 5: consumers compute nothing;
the last two give invalid records): what the ray costs, what the hand-off costs.  usage: perf_trio_diag.py [TRIO values ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geoac_amd as G
import harness as H
vals = sys.argv[1:] or ["0", "1", "3", "5"]
th = np.full(64, 0.5); ph = -180.0 + 360.0 * np.arange(64) / 64
for trio in vals + vals:
    ctx = G.FanContext(G.EQ_GLOBAL, device=0, options={"TRIO": trio}); ctx.load_met(H.TOYATMO)
    ctx.set_params(bounces=2, calc_amp=1, mode=0, src=(0.0, 30.0, 0.0))
    ctx.set_angles(th, ph); ctx.launch()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); ctx.launch(); ts.append(time.perf_counter() - t0)
    print(f"TRIO={trio}: {min(ts) * 1e3:7.2f} ms -> {min(ts) / 54130 * 1e6:.3f} us per step of the longest ray (54 130 steps)", flush=True)
    ctx.close()
