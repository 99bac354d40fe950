"""A/B of builds of libgeoac_hip.so on the metric fan, same box, same process, each build twice in turn: usage ab_metric.py <passes> <libA.so> <libB.so> [...]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geoac_amd.api as A
import harness as H


def run(lib_path, passes):
    A._lib = None
    A.library_path = lambda: lib_path
    import geoac_amd as G
    which = os.environ.get("GEOAC_AB_SET", "")
    if which == "cfg3":
        th, ph = A.fan_enumerate(theta_min=0.25, theta_max=45.0, theta_step=0.25, phi_min=-180.0, phi_max=179.5, phi_step=0.5)
        ctx = A.FanContext(A.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO); ctx.set_params(bounces=3, calc_amp=1, mode=0, src=(0.0, 30.0, 0.0))
    else:
        th, ph = A.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0)
        ctx = A.FanContext(A.EQ_3D if which == "3d" else A.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO); ctx.set_params(bounces=2, calc_amp=1, mode=0)
    ctx.set_angles(th, ph); ctx.launch(); ctx.launch()
    ts = []
    for _ in range(passes):
        t0 = time.perf_counter(); ctx.launch(); ts.append((time.perf_counter() - t0) * 1e3)
    tm = ctx.timing()
    print(os.path.dirname(lib_path).split("/")[-1] + "/" + os.path.basename(lib_path), os.environ.get("GEOAC_AB_TAG", ""), "ms per pass: min %.2f median %.2f" % (min(ts), float(np.median(ts))),
          "rk4 %.2f post %.2f epochs %d" % (tm["ms_rk4"], tm["ms_post"], tm["epochs"]), flush=True)
    ctx.close()


if __name__ == "__main__":
    n = int(sys.argv[1])
    for p in sys.argv[2:] + sys.argv[2:]:
        run(os.path.abspath(p), n)
