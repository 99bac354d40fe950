"""config 5 of SURVEY §8d: -eig_search to 64 receivers on a 2.5-degree ring around the source, bounces 0..2.
usage: perf_eigenray.py [global|globalrd] [n_rcvr]"""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geoac_amd as G
import harness as H

which = sys.argv[1] if len(sys.argv) > 1 else "global"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if which == "globalrd":
    import rngdep_data as RD
    grid = RD.write_grid_global(os.path.join(tempfile.gettempdir(), "gge"), short_paths=False)
    ctx = G.FanContext(G.EQ_GLOBAL_RNGDEP, device=0); ctx.load_grid(*grid); lat0, lon0 = 31.0, 0.0
else:
    ctx = G.FanContext(G.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO); lat0, lon0 = 30.0, 0.0
ctx.set_params(src=(0.0, lat0, lon0))
az = np.arange(n) * (2.0 * np.pi / n)
rcv = np.stack([lat0 + 2.5 * np.cos(az), lon0 + 2.5 * np.sin(az) / np.cos(np.radians(lat0))], axis=1)
ctx.eig_search(rcv[:2], bnc_min=0, bnc_max=0)          # warm-up
t0 = time.perf_counter(); out = ctx.eig_search(rcv, bnc_min=0, bnc_max=2); dt = time.perf_counter() - t0
st = out["stats"]
print(f"{which}: {n} receivers, bounces 0..2: {len(out['eig'])} eigenrays in {dt:.2f} s; {st['rays']} rays, {st['steps']} RK4 ray-steps, "
      f"{st['launches']} fan launches in {st['rounds']} rounds; {st['steps'] / dt:.3e} ray-steps/s")
