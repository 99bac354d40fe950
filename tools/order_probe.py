"""Config-4 share with the rays of a wave taken along the azimuth instead of along the inclination (same rays, permuted): time per pass and,
with a -DGEOAC_KSTAT build in GEOAC_LIB, the distinct (segment, cell) keys per wave-stage.  usage: order_probe.py [thin] [orders...]"""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geoac_amd as G
import rngdep_data as RD

thin = int(sys.argv[1]) if len(sys.argv) > 1 else 1
orders = sys.argv[2:] or ["incl", "az", "az8"]
grid = RD.write_grid(os.path.join(tempfile.gettempdir(), f"gdb{thin}"), short_paths=False, thin=thin)
with G.options(SORT=int(os.environ.get("ORDER_PROBE_SORT", "0"))):          # 0: the order given here is the order of the lanes
    ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0)
ctx.load_grid(*grid)
ctx.set_params(bounces=1, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
th, ph = G.fan_enumerate(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 124 * 0.36, phi_step=0.36)
naz = len(np.unique(ph)); nin = len(th) // naz; assert naz * nin == len(th) and np.all(ph[:nin] == ph[0])
idx = np.arange(len(th)).reshape(naz, nin)                    # [az][incl], incl fast (the reference's loop order)
ref = None
for o in orders:
    if o == "incl": perm = idx.reshape(-1)
    elif o == "az": perm = idx.T.reshape(-1)                  # azimuth fast
    elif o.startswith("az"):                                  # tiles of b inclinations x (64 / b) azimuths per wave
        b = int(o[2:]); a = 64 // b
        naz_p = (naz + a - 1) // a * a; nin_p = (nin + b - 1) // b * b
        big = -np.ones((naz_p, nin_p), dtype=np.int64); big[:naz, :nin] = idx
        t = big.reshape(naz_p // a, a, nin_p // b, b).transpose(0, 2, 1, 3).reshape(-1)
        perm = t[t >= 0]
    ctx.set_angles(th[perm], ph[perm]); ctx.launch()
    ts = []
    for _ in range(2):
        t0 = time.perf_counter(); ctx.launch(); ts.append(time.perf_counter() - t0)
    rec, _ = ctx.fetch()
    back = np.empty_like(rec); back[perm] = rec
    if ref is None: ref = back
    same = np.array_equal(ref.view(np.uint64), back.view(np.uint64))
    tm = ctx.timing()
    print(f"order {o}: {min(ts)*1e3:.1f} ms per pass, steps {ctx.total_steps()}, {ctx.total_steps()/min(ts):.3e} steps/s, rk4 {tm['ms_rk4']:.1f} ms, records equal to the first order's: {same}", flush=True)
