"""set-up time of a range-dependent atmosphere: geoac_upload_atmo_3d with the table built on the device (default) vs on the host
(GEOAC_GRID_BUILD=host).  usage: perf_gridbuild.py nx ny nz [host]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geoac_amd as G
from test_gpu_gridbuild import _synthetic_grid

nx, ny, nz = (int(a) for a in sys.argv[1:4])
host = len(sys.argv) > 4 and sys.argv[4] == "host"
if host:
    G.DEFAULT_OPTIONS["GRID_BUILD"] = "host"
x, y, z, T, u, v, rho = _synthetic_grid(nx, ny, nz, False)
ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0)
ctx.upload_atmo_3d(x[:2], y[:2], z[:3], T[:2, :2, :3].copy(), u[:2, :2, :3].copy(), v[:2, :2, :3].copy(), rho[:2, :2, :3].copy())   # warm-up: context, kernels
t0 = time.perf_counter()
ctx.upload_atmo_3d(x, y, z, T, u, v, rho)
dt = time.perf_counter() - t0
tab_gb = (3 * 40 + 16) * (nz - 1) * nx * ny * 8 / 1e9
print(f"{nx}x{ny}x{nz} grid ({'host' if host else 'device'} builder): {12 * nx * ny} spline systems of {nz} unknowns, table {tab_gb:.2f} GB, upload_atmo_3d {dt:.3f} s")
ctx.close()
