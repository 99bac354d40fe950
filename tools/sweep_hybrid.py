"""metric fan under several (PAIR_FRAC, HYBRID_ROWS) launch plans in ONE process on one box (box-to-box differences are ~2 %): min / median ms per pass.
usage: sweep_hybrid.py [passes] [global|3d] [pf:hr ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geoac_amd as G
import harness as H
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 5
which = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] in ("global", "3d") else "global"
plans = [a for a in sys.argv[2:] if ":" in a] or ["0.10:0.70", "0.10:0.75", "0.10:0.80", "0.15:0.75", "0.08:0.75", "0.10:0.75"]
th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0)
for pl in plans:
    pf, hr = pl.split(":")
    ctx = G.FanContext(G.EQ_GLOBAL if which == "global" else G.EQ_3D, device=0, options={"PAIR_FRAC": pf, "HYBRID_ROWS": hr}); ctx.load_met(H.TOYATMO)
    ctx.set_params(bounces=2, calc_amp=1, mode=0)
    ctx.set_angles(th, ph); ctx.launch(); ctx.launch()
    ts = []
    for _ in range(passes):
        t0 = time.perf_counter(); ctx.launch(); ts.append((time.perf_counter() - t0) * 1e3)
    tm = ctx.timing()
    print(f"pair_frac {pf} hybrid_rows {hr}: min {min(ts):.2f} median {float(np.median(ts)):.2f} ms per pass, rk4 {tm['ms_rk4']:.1f}, epochs {tm['epochs']}", flush=True)
    ctx.close()
