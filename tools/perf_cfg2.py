"""config 2 (GeoAc3D 360 x 90, CalcAmp) on the library's own stream: time per pass and the RK4 / post-pass event sums"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geoac_amd as G
import harness as H
ctx = G.FanContext(G.EQ_3D, device=0); ctx.load_met(H.TOYATMO)
ctx.set_params(bounces=2, calc_amp=1, mode=0)
th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0)
ctx.set_angles(th, ph); ctx.launch()
for _ in range(2):
    t0 = time.perf_counter(); ctx.launch(); dt = time.perf_counter() - t0
    tm = ctx.timing()
    print(f"{ctx.total_steps()} steps, {dt*1e3:.1f} ms, {ctx.total_steps()/dt:.3e} steps/s (rk4 {tm['ms_rk4']:.1f} ms, post {tm['ms_post']:.1f} ms, epochs {tm['epochs']})")
rec, _ = ctx.fetch()
print("longest ray:", int(rec[:, :, 1].sum(axis=1).max()), "steps")
