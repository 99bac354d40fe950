cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
python - > gpurun_out/r04e/cfg4_eik.log 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, tempfile
import geoac_amd as G
import rngdep_data as RD
grid = RD.write_grid(os.path.join(tempfile.gettempdir(), "g4"), short_paths=False, thin=1)
th, ph = G.fan_enumerate(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 124 * 0.36, phi_step=0.36)
ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0); ctx.load_grid(*grid); ctx.set_params(bounces=1, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
rec, steps = ctx.run(th, ph)
REC = G.REC
valid = rec[..., REC["VALID"]] > 0
st = rec[..., REC["STATE"]:REC["STATE"] + 6][valid]
_, a7 = ctx.probe_grid(st[:, 0], st[:, 1], st[:, 2]); _, a0 = ctx.probe_grid(np.zeros(1), np.zeros(1), np.zeros(1))
nu = np.sqrt((st[:, 3:6] ** 2).sum(axis=1))
# with the wind term: |nu| = c0 / c (1 - nu . wind / c0)  ->  H = |nu| - c0/c + nu.wind/c
u, v = a7[:, 2], a7[:, 3]
H0 = nu * a7[:, 0] / a0[0, 0] - 1.0
H1 = nu - a0[0, 0] / a7[:, 0] + (st[:, 3] * u + st[:, 4] * v) / a7[:, 0]
thv = np.repeat(th[:, None], 2, axis=1)[valid]
for name, e in (("|nu| c / c0 - 1", np.abs(H0)), ("Hamiltonian with the wind term", np.abs(H1))):
    print(name, "max %.3e  99.9%% %.3e  99%% %.3e  median %.3e" % (e.max(), np.quantile(e, 0.999), np.quantile(e, 0.99), np.median(e)))
    w = np.argsort(e)[-8:]
    print("  worst at theta", thv[w], "z", st[w, 2], "u,v", u[w], v[w], "err", e[w])
PY
cat gpurun_out/r04e/cfg4_eik.log
