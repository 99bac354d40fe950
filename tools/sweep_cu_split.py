"""CU partition between the RK4 launches and the post-pass (option CU_SPLIT = compute units set aside for the post-pass; streams with CU masks) on the fans that fill the chip:
ms per pass and records bit for bit against the unpartitioned plan.  usage: sweep_cu_split.py [cfg3|cfg2|metric] [passes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geoac_amd as G
import harness as H
which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if which == "cfg3":
    th, ph = G.fan_enumerate(theta_min=0.25, theta_max=45.0, theta_step=0.25, phi_min=-180.0, phi_max=179.5, phi_step=0.5); params = dict(bounces=3, calc_amp=1, mode=0, src=(0.0, 30.0, 0.0))
elif which == "g48k":
    th, ph = G.fan_enumerate(theta_min=0.5, theta_max=45.0, theta_step=0.5, phi_min=-180.0, phi_max=179.4, phi_step=0.675); params = dict(bounces=2, calc_amp=1, mode=0, src=(0.0, 30.0, 0.0))
elif which == "g72k":
    th, ph = G.fan_enumerate(theta_min=0.25, theta_max=45.0, theta_step=0.25, phi_min=-180.0, phi_max=179.1, phi_step=0.9); params = dict(bounces=3, calc_amp=1, mode=0, src=(0.0, 30.0, 0.0))
elif which == "g200k":
    th, ph = G.fan_enumerate(theta_min=0.2, theta_max=50.0, theta_step=0.2, phi_min=-180.0, phi_max=179.55, phi_step=0.45); params = dict(bounces=2, calc_amp=1, mode=0, src=(0.0, 30.0, 0.0))
elif which == "3dmetric":        # BASELINE config 2: GeoAc3D 360 az x 90 incl, 2 bounces
    th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0); params = dict(bounces=2, calc_amp=1, mode=0)
elif which == "3dbig":
    th, ph = G.fan_enumerate(theta_min=0.25, theta_max=45.0, theta_step=0.25, phi_min=-180.0, phi_max=179.5, phi_step=0.5); params = dict(bounces=3, calc_amp=1, mode=0)
elif which == "cfg2":
    th, ph = G.fan_enumerate(theta_min=0.5, theta_max=45.0, theta_step=0.5, phi_min=-180.0, phi_max=179.0, phi_step=1.0); params = dict(bounces=10, calc_amp=1, mode=0, src=(0.0, 30.0, 0.0))
else:
    th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0); params = dict(bounces=2, calc_amp=1, mode=0)
ref = None
plans = [{}] + [{"CU_SPLIT": str(n)} for n in (48, 64, 72, 80, 88, 96, 112)] + [{"CU_SPLIT": str(n), "PP_LDS_TABLE": "1"} for n in (64, 80, 96)] + [{}]
if len(sys.argv) > 3:
    plans = [{}] + [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[3:]] + [{}]
for opts in plans:
    ctx = G.FanContext(G.EQ_3D if which in ("3dbig", "3dmetric") else G.EQ_GLOBAL, device=0, options=opts); ctx.load_met(H.TOYATMO); ctx.set_params(**params)
    ctx.set_angles(th, ph); ctx.launch()
    ts = []
    for _ in range(passes):
        t0 = time.perf_counter(); ctx.launch(); ts.append((time.perf_counter() - t0) * 1e3)
    rec, steps = ctx.fetch(); tm = ctx.timing()
    if ref is None:
        ref = rec.copy()
    print(which, opts, "ms per pass min %.1f median %.1f" % (min(ts), float(np.median(ts))), "rk4 %.1f post %.1f epochs %d" % (tm["ms_rk4"], tm["ms_post"], tm["epochs"]),
          "-> %.3e ray-steps/s" % (steps / (min(ts) * 1e-3)), "bit-identical:", bool(np.array_equal(ref.view(np.uint64), rec.view(np.uint64))), flush=True)
    ctx.close()
