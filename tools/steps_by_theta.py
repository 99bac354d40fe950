import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, geoac_amd as G, harness as H
ctx = G.FanContext(G.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO)
ctx.set_params(bounces=2, calc_amp=1, mode=0, src=(0.0, 30.0, 0.0))
th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0)
rec, steps = ctx.run(th, ph)
s = rec[:, :, 1].sum(axis=1).reshape(360, 90)
print("theta  max  mean  min over azimuth")
for j in range(0, 90, 4):
    print(f"{th[j]:5.1f} {int(s[:, j].max()):6d} {int(s[:, j].mean()):6d} {int(s[:, j].min()):6d}")
srt = np.sort(s.max(axis=0))[::-1]
print("sorted max-steps by theta:", srt[:12].astype(int), "...", srt[-5:].astype(int))
