"""the table post-pass per path segment as a function of the launch inclination (NO_OVERLAP=1: RK4 and post-pass in turn, so the post-pass's event span is its own time):
shallow rays change spline segment every tens of steps, steep ones every two to four - is the post-pass paying for the table gathers?  usage: perf_postpass_by_incl.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geoac_amd as G
import harness as H
for lo, hi in ((0.25, 5.0), (5.25, 10.0), (15.25, 20.0), (30.25, 35.0), (40.25, 45.0), (60.25, 65.0), (80.25, 85.0)):
    th, ph = G.fan_enumerate(theta_min=lo, theta_max=hi, theta_step=0.25, phi_min=-180.0, phi_max=179.9, phi_step=0.1)      # 20 x 3600 = 72 000 rays
    for opts in ({"NO_OVERLAP": "1"}, {"NO_OVERLAP": "1", "PP_ONETRIP": "0"}):
        ctx = G.FanContext(G.EQ_GLOBAL, device=0, options=opts); ctx.load_met(H.TOYATMO); ctx.set_params(bounces=1, calc_amp=1, mode=0, src=(0.0, 30.0, 0.0))
        ctx.set_angles(th, ph); ctx.launch(); ctx.launch()
        rec, steps = ctx.fetch(); tm = ctx.timing()
        print("inclinations %.2f-%.2f, %d rays, %s: %d steps, rk4 %.1f ms (%.3f ns per ray-step), post-pass + sums %.1f ms (%.3f ns per segment)"
              % (lo, hi, len(th), opts, steps, tm["ms_rk4"], tm["ms_rk4"] * 1e6 / steps, tm["ms_post"], tm["ms_post"] * 1e6 / steps), flush=True)
        ctx.close()
