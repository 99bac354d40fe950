#!/usr/bin/env python3
"""GPU box: integrate the metric, config-2 and config-3 fans and LIST the arrivals whose amplitude differs from the compiled reference's
by more than 1e-6 (row of the fixture's value table, leg, error, 4 x the reference's own sensitivity) -> gpurun_out/amp_loose.json.
`python tests/golden/make_golden_full.py exempt gpurun_out/amp_loose.json` (container: needs nothing but the fixtures) then admits the ones
the reference's sensitivity covers into the fixtures' `amp_exempt` list; tests/parity.py fails on any loose arrival that is not named."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geoac_amd as G  # noqa: E402
import harness as H  # noqa: E402
from parity import compare_compact  # noqa: E402

out = {}
for name, eq in (("metric", G.EQ_GLOBAL), ("cfg2", G.EQ_3D), ("cfg3", G.EQ_GLOBAL)):
    g = np.load(os.path.join(H.GOLDEN_DIR, f"full_{name}.npz"))
    th, ph = G.fan_enumerate(**{str(k): float(v) for k, v in g["fan"]})
    ctx = G.FanContext(eq, device=0)
    ctx.load_met(H.TOYATMO)
    ctx.set_params(bounces=int(g["bounces"]), calc_amp=1, mode=0)
    rec, steps = ctx.run(th, ph)
    assert steps == int(g["total_steps"])
    err = compare_compact(rec, g, collect_loose=True)
    out[name] = err["AMP_exempt"]
    print(name, len(err["AMP_exempt"]), "arrivals beyond 1e-6:", err["AMP_exempt"], flush=True)
    ctx.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "amp_loose.json"), "w"))
