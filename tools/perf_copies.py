"""where the milliseconds of a timed metric pass go besides the kernels: geoac_fan_set_angles (host sort of the rays by inclination, permutation and angles host -> device),
geoac_fan_launch, geoac_fan_fetch (records device -> host, pageable and pinned destination).  usage: perf_copies.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import geoac_amd as G
import harness as H
th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0)
ctx = G.FanContext(G.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO); ctx.set_params(bounces=2, calc_amp=1, mode=0)
pinned = torch.empty((len(th), 3, G.REC_STRIDE), dtype=torch.float64, pin_memory=True).numpy()
pageable = np.empty((len(th), 3, G.REC_STRIDE))
ctx.run(th, ph); ctx.run(th, ph, out=pinned)
acc = {"set_angles": [], "launch": [], "fetch_pinned": [], "fetch_pageable": []}
for _ in range(6):
    t0 = time.perf_counter(); ctx.set_angles(th, ph); t1 = time.perf_counter(); ctx.launch(); t2 = time.perf_counter(); ctx.fetch(pinned); t3 = time.perf_counter(); ctx.fetch(pageable); t4 = time.perf_counter()
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
        acc[k].append(v * 1e3)
print({k: "min %.3f median %.3f ms" % (min(v), float(np.median(v))) for k, v in acc.items()}, ctx.timing())
