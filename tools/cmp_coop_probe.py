import os, sys, tempfile, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import geoac_amd as G, harness as H, rngdep_data as RD
grid = RD.write_grid(tempfile.mkdtemp(), short_paths=False, thin=1)
ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0); ctx.load_grid(*grid)
ctx.set_params(bounces=1, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
th, ph = G.fan_enumerate(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 3 * 0.36, phi_step=0.36)
rec, st = ctx.run(th, ph)
rng = np.random.default_rng(1)
n = 4096
x = rng.uniform(-400, 400, n); y = rng.uniform(-400, 400, n); z = rng.uniform(0.0, 130.0, n)
o0, a0 = ctx.probe_grid(x, y, z, coop=False)
o1, a1 = ctx.probe_grid(x, y, z, coop=True)
d = np.abs(o1 - o0); sc = np.abs(o0).max(axis=0) + 1e-300
print("probe: differing entries", int((o1 != o0).sum()), "of", o0.size, "max abs diff / column scale", (d / sc).max())
bad = np.argwhere(o1 != o0)
print("columns that differ:", sorted(set(bad[:, 1].tolist()))[:40])
pts = np.unique(bad[:, 0])
print("points that differ:", len(pts), "of", n)
rel = (d / sc).max(axis=1)
worst = np.argsort(rel)[-8:]
for i in worst: print("  pt", i, "x,y,z", x[i], y[i], z[i], "rel", rel[i], "cols", np.nonzero(o1[i] != o0[i])[0][:12])
gx, gy = np.loadtxt(grid[1]) if False else (None, None)
inside = (np.abs(x) <= 300) & (np.abs(y) <= 300)
print("differing inside |x|,|y|<=300:", int(np.isin(np.nonzero(inside)[0], pts).sum()), "of", int(inside.sum()))
# same evaluation twice (race check): coop twice
o2, _ = ctx.probe_grid(x, y, z, coop=True)
print("coop run twice identical:", bool(np.array_equal(o1, o2)))
o3, _ = ctx.probe_grid(x, y, z, coop=False)
print("per-lane run twice identical:", bool(np.array_equal(o0, o3)))
np.set_printoptions(precision=17, linewidth=200)
for c in (0, 1, 4, 6, 16, 27):
    m = o1[:, c] != o0[:, c]
    if m.any():
        r = np.abs(o1[m, c] - o0[m, c]) / np.maximum(np.abs(o0[m, c]), 1e-300)
        print("col", c, "differs at", int(m.sum()), "points; max rel-to-value", r.max(), "median", np.median(r), "col scale", sc[c], " example", o0[m, c][:2], o1[m, c][:2])
