"""the wave-specialised kernel (TRIO=1: k_rk4_trio, the ray on one wave, one launch-angle system on each of two more) against the default plan:
records bit for bit on small fans, then the step time of one wave of shallow rays and the metric fan.  usage: perf_trio.py [quick]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geoac_amd as G
import harness as H


def fan(th, ph, opts, bounces=2, passes=0, src=(0.0, 30.0, 0.0)):
    ctx = G.FanContext(G.EQ_GLOBAL, device=0, options=opts); ctx.load_met(H.TOYATMO)
    ctx.set_params(bounces=bounces, calc_amp=1, mode=0, src=src)
    ctx.set_angles(th, ph); ctx.launch()
    ts = []
    for _ in range(passes):
        t0 = time.perf_counter(); ctx.launch(); ts.append(time.perf_counter() - t0)
    rec, steps = ctx.fetch()
    tm = ctx.timing()
    ctx.close()
    return rec.copy(), steps, (min(ts) if ts else 0.0), tm


ok = True
for name, th, ph in (("64 x 0.5 deg", np.full(64, 0.5), -180.0 + 360.0 * np.arange(64) / 64),
                     ("100 mixed", np.linspace(0.5, 45.0, 100), np.linspace(-170.0, 170.0, 100)),
                     ("7 rays", np.array([1.0, 3.0, 8.0, 15.0, 22.0, 30.0, 41.0]), np.array([-90.0, -45.0, 0.0, 10.0, 45.0, 90.0, 180.0])),
                     ("300 steep", np.linspace(20.0, 89.0, 300), np.linspace(-180.0, 179.0, 300))):
    a = fan(th, ph, {"TRIO": "0"})
    for extra in ({}, {"S_ROWS": "512"}):
        b = fan(th, ph, dict({"TRIO": "1"}, **extra))
        same = a[1] == b[1] and np.array_equal(a[0].view(np.uint64), b[0].view(np.uint64))
        if not same:
            d = np.argwhere(a[0].view(np.uint64) != b[0].view(np.uint64))
            print("   first differences (ray, leg, field):", d[:8].tolist(), "steps", a[1], b[1])
        ok &= same
        print(f"{name:14s} {extra}: steps {a[1]} / {b[1]}, records bit-identical: {same}", flush=True)
if not ok:
    sys.exit("TRIO records differ")
if "quick" in sys.argv:
    sys.exit(0)
for theta in (0.5, 2.0):
    for n_az in (64, 128):
        th = np.full(n_az, theta); ph = -180.0 + 360.0 * np.arange(n_az) / n_az
        for trio in ("0", "1", "0", "1"):
            rec, steps, t, tm = fan(th, ph, {"TRIO": trio}, passes=3)
            longest = rec[:, :, 1].sum(axis=1).max()
            print(f"theta {theta:4.1f} {n_az:3d} rays TRIO={trio}: {t * 1e3:7.2f} ms, longest ray {int(longest)} steps -> {t / longest * 1e6:.3f} us per step (epochs {tm['epochs']})", flush=True)
th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0)
ref = None
for opts in ({"TRIO": "0"}, {"TRIO": "1", "TRACE_EPOCHS": "1"}, {"TRIO": "0"}, {"TRIO": "1"}, {"TRIO": "1", "PAIR_FRAC": "0.2"}, {"TRIO": "1", "PAIR_FRAC": "0.3"}):
    rec, steps, t, tm = fan(th, ph, opts, passes=4)
    if ref is None:
        ref = rec
    print("metric fan", opts, "ms per pass %.2f" % (t * 1e3), "rk4 %.2f post %.2f epochs %d" % (tm["ms_rk4"], tm["ms_post"], tm["epochs"]),
          "bit-identical to the first:", np.array_equal(ref.view(np.uint64), rec.view(np.uint64)), flush=True)
