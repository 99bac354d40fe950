"""sha1 of the records of the config-4 share (and of a small spherical grid fan): bitwise A/B of builds.  usage: rec_checksum.py"""
import hashlib, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geoac_amd as G, harness as H, rngdep_data as RD
grid = RD.write_grid(tempfile.mkdtemp(), short_paths=False, thin=1)
ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0); ctx.load_grid(*grid)
ctx.set_params(bounces=1, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
th, ph = G.fan_enumerate(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 124 * 0.36, phi_step=0.36)
rec, st = ctx.run(th, ph)
print("cfg4 share", st, hashlib.sha1(np.ascontiguousarray(rec).tobytes()).hexdigest(), "sum ttime", repr(float(rec[..., H.REC["TTIME"]].sum())), "sum atten", repr(float(rec[..., H.REC["ATTEN"]].sum())))
