"""Integrates a named full-size fan on the GPU and saves its record table to gpurun_out/ (diagnostics; compare offline with the fixtures).
usage: dump_fan.py metric|cfg2|cfg3"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geoac_amd as G
import harness as H
which = sys.argv[1]
g = np.load(os.path.join(H.GOLDEN_DIR, f"full_{which}.npz"))
th, ph = G.fan_enumerate(**{str(k): float(v) for k, v in g["fan"]})
ctx = G.FanContext(G.EQ_3D if which == "cfg2" else G.EQ_GLOBAL, device=0)
ctx.load_met(H.TOYATMO); ctx.set_params(bounces=int(g["bounces"]), calc_amp=1, mode=0)
rec, steps = ctx.run(th, ph)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", f"rec_{which}.npy"), rec[..., :12].astype(np.float64))
print(which, steps)
