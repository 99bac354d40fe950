"""How full are the waves of a fan?  Integrates a configuration, takes the per-ray step counts from the records and reports, for the
integration order (inclination-sorted slots, 64 per wave): sum(steps) / (64 * sum over waves of the longest ray of the wave) - the lane
efficiency of a schedule without compaction - and the same with compaction at epoch boundaries of `rows` steps.
usage: lane_efficiency.py cfg4|cfg3|cfg2|metric [rows]"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geoac_amd as G
import harness as H


def fan(which):
    if which == "cfg4":
        import rngdep_data as RD
        grid = RD.write_grid(os.path.join(tempfile.gettempdir(), "gdl"), short_paths=False, thin=1)
        ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0); ctx.load_grid(*grid)
        ctx.set_params(bounces=1, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
        th, ph = G.fan_enumerate(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 124 * 0.36, phi_step=0.36)
    elif which == "cfg3":
        ctx = G.FanContext(G.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO); ctx.set_params(bounces=3, calc_amp=1, mode=0)
        th, ph = G.fan_enumerate(theta_min=0.25, theta_max=45.0, theta_step=0.25, phi_min=-180.0, phi_max=179.5, phi_step=0.5)
    else:
        ctx = G.FanContext(G.EQ_3D if which == "cfg2" else G.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO); ctx.set_params(bounces=2, calc_amp=1, mode=0)
        th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0)
    rec, steps = ctx.run(th, ph)
    return th, ph, rec[..., H.REC["STEPS"]].sum(axis=1).astype(np.int64), steps


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    th, ph, st, total = fan(which)
    order = np.argsort(th, kind="stable")
    s = st[order]
    pad = (-len(s)) % 64
    w = np.concatenate([s, np.zeros(pad, dtype=s.dtype)]).reshape(-1, 64)
    print(f"{which}: {len(s)} rays, {total} ray-steps; steps per ray min {s.min()} mean {s.mean():.0f} max {s.max()}")
    print(f"  no compaction: lane efficiency {s.sum() / (64.0 * w.max(axis=1).sum()):.3f}  ({w.shape[0]} waves)")
    # compaction at epoch boundaries: in epoch e the live rays (steps > e*rows) are packed into ceil(live/64) waves, each running min(rows, remaining)
    e, work = 0, 0.0
    while True:
        rem = s - e * rows
        live = rem[rem > 0]
        if live.size == 0:
            break
        lw = np.sort(np.minimum(live, rows))[::-1]
        padl = (-len(lw)) % 64
        lw = np.concatenate([lw, np.zeros(padl, dtype=lw.dtype)]).reshape(-1, 64)
        work += 64.0 * lw.max(axis=1).sum()
        e += 1
    print(f"  compaction every {rows} steps ({e} epochs, live rays packed and ordered by remaining length): {s.sum() / work:.3f}")
    for r2 in (rows * 2, rows * 4):
        e, work = 0, 0.0
        while True:
            rem = s - e * r2
            live = rem[rem > 0]
            if live.size == 0: break
            lw = np.minimum(live, r2)          # keep slot order (no sort): packed but unsorted
            padl = (-len(lw)) % 64
            lw = np.concatenate([lw, np.zeros(padl, dtype=lw.dtype)]).reshape(-1, 64)
            work += 64.0 * lw.max(axis=1).sum(); e += 1
        print(f"  compaction every {r2} steps, slot order kept: {s.sum() / work:.3f}")


if __name__ == "__main__":
    main()
