"""how long the serial chain of the critical ray is when only the base ray is integrated (calc_amp=0: no launch-angle systems) - what a wave-specialised kernel
(base ray on one SIMD, the launch-angle systems on others) would be left with.  Whole metric fan, and one wave of one inclination.  usage: perf_base_chain.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geoac_amd as G
import harness as H
th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0)
for amp in (1, 0, 1, 0):
    ctx = G.FanContext(G.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO); ctx.set_params(bounces=2, calc_amp=amp, mode=0)
    ctx.set_angles(th, ph); ctx.launch(); ctx.launch()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.launch(); ts.append((time.perf_counter() - t0) * 1e3)
    tm = ctx.timing()
    print("whole fan calc_amp", amp, "ms per pass min %.2f median %.2f" % (min(ts), float(np.median(ts))), "rk4 %.2f post %.2f epochs %d" % (tm["ms_rk4"], tm["ms_post"], tm["epochs"]), flush=True)
    ctx.close()
for theta in (0.5, 2.0):
    for amp in (1, 0):
        for n_az in (32, 64):
            ctx = G.FanContext(G.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO)
            ctx.set_params(bounces=2, calc_amp=amp, mode=0, src=(0.0, 30.0, 0.0))
            p = -180.0 + 360.0 * np.arange(n_az) / n_az
            t = np.full(n_az, theta)
            ctx.set_angles(t, p); ctx.launch()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); ctx.launch(); ts.append(time.perf_counter() - t0)
            rec, steps = ctx.fetch()
            longest = rec[:, :, 1].sum(axis=1).max()
            tm = ctx.timing()
            print(f"theta {theta:4.1f} calc_amp {amp} {n_az:3d} rays: {min(ts) * 1e3:7.2f} ms, longest ray {int(longest)} steps -> {min(ts) / longest * 1e6:.3f} us per step (rk4 {tm['ms_rk4']:.1f} ms, epochs {tm['epochs']})", flush=True)
            ctx.close()
