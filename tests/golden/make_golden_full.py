"""Full-size fixtures of BASELINE.json's configurations, made by the COMPILED, UNMODIFIED reference
(oracle/_ref/libref_*.so, built by oracle/Makefile from /root/reference).  Run here only - the reference does not travel:

    make -C oracle all && python tests/golden/make_golden_full.py [metric cfg2 cfg3 cfg4 cfg5] [-j 6]

One worker process per core (the reference keeps its state in globals: one equation set and one atmosphere per process); every
worker integrates whole azimuths with ref_fan, i.e. the reference's own GeoAc_Propagate_RK4 / post-pass in the order its *_RunProp
loops call them.  What is kept per (ray, leg) is compact - data, not code:

    steps  int32   GeoAc_Propagate_RK4's return value (0 = leg not run)
    flags  int8    bit 0: a results row is written (VALID), bit 1: BreakCheck ended the leg (BROKE)
    vals   float64 TTIME, ATTEN, TURN, INCL, BACKAZ, AMP, RANGE (the columns of a _results.dat row, unformatted)

    sens   float32 (second pass, `sens_<name>`) conditioning of the reference's OWN amplitude: |AMP(theta (1 + 1e-12)) - AMP(theta)| / |AMP|,
                   the same reference binary run again with the launch inclination changed in its 12th digit.  A handful of arrivals
                   (rays trapped in a duct, Jacobian 1e3-1e4 x the typical one) answer a 1e-14 change of theta with a 1e-6 change
                   of amplitude: no arithmetic other than the reference's own bit pattern can match those to 1e-6, so the parity
                   tests judge AMP by max(1e-6, 4 x sens) and report how many arrivals needed the second bound.

`vals` is kept for every `vals_every`-th azimuth (all of them for the metric fan and config 2; every 4th for config 3).
tests/test_gpu_fullfan.py compares every ray of the GPU fans with these: counts exact, values within 1e-6 relative.
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import harness as H  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
VAL_FIELDS = ("TTIME", "ATTEN", "TURN", "INCL", "BACKAZ", "AMP", "RANGE")

# the fans, exactly as tools/bench_configs.py / bench.py / tests enumerate them
FANS = {
    "metric": dict(eq=H.EQ_GLOBAL, fan=dict(phi_min=-180.0, phi_max=179.0, phi_step=1.0), bounces=2, vals_every=1),
    "cfg2": dict(eq=H.EQ_3D, fan=dict(phi_min=-180.0, phi_max=179.0, phi_step=1.0), bounces=2, vals_every=1),
    "cfg3": dict(eq=H.EQ_GLOBAL, fan=dict(theta_min=0.25, theta_max=45.0, theta_step=0.25, phi_min=-180.0, phi_max=179.5, phi_step=0.5),
                 bounces=3, vals_every=4),
}

_shim = {}


def _work(job):
    name, eq, grid, cfgkw, th, ph = job[:6]
    if len(job) > 6:
        th = th * (1.0 + job[6])
    key = (eq, grid)
    if key not in _shim:
        _shim[key] = H.RefShim(eq, grid=grid) if grid else H.RefShim(eq)
    cfg = H.make_cfg(eq, **cfgkw)
    steps, rec, _, _ = _shim[key].fan(cfg, th, ph)
    return compact(rec), steps


def compact(rec):
    st = rec[..., H.REC["STEPS"]].astype(np.int32)
    fl = ((rec[..., H.REC["VALID"]] > 0).astype(np.int8) | ((rec[..., H.REC["BROKE"]] > 0).astype(np.int8) << 1))
    vals = np.stack([rec[..., H.REC[f]] for f in VAL_FIELDS], axis=-1)
    return st, fl, vals


def run_jobs(pool, jobs, label):
    t0 = time.time()
    out = []
    for i, r in enumerate(pool.imap(_work, jobs, chunksize=1)):
        out.append(r)
        if (i + 1) % 20 == 0 or i + 1 == len(jobs):
            print(f"  {label}: {i + 1}/{len(jobs)} jobs, {time.time() - t0:.0f} s", flush=True)
    return out


def stratified(pool, name):
    spec = FANS[name]
    th, ph = H.fan_angles(**spec["fan"])
    # one job per azimuth (phi outer / theta inner in the enumeration)
    edges = np.flatnonzero(np.diff(ph) != 0) + 1
    groups = np.split(np.arange(len(th)), edges)
    cfgkw = dict(bounces=spec["bounces"], calc_amp=True, mode=0)
    jobs = [(name, spec["eq"], None, cfgkw, th[g], ph[g]) for g in groups]
    res = run_jobs(pool, jobs, name)
    steps = np.concatenate([r[0][0] for r in res]); flags = np.concatenate([r[0][1] for r in res])
    ev = spec["vals_every"]
    vsel = np.concatenate([g for i, g in enumerate(groups) if i % ev == 0])
    vals = np.concatenate([r[0][2] for i, r in enumerate(res) if i % ev == 0])
    total = int(sum(r[1] for r in res))
    assert total == int(steps.sum())
    path = os.path.join(OUT, f"full_{name}.npz")
    np.savez_compressed(path, fan=np.array(sorted(spec["fan"].items()), dtype=object).astype(str), bounces=spec["bounces"],
                        n_rays=len(th), steps=steps, flags=flags, vals_idx=vsel.astype(np.int32), vals=vals,
                        val_fields=np.array(VAL_FIELDS), total_steps=np.int64(total))
    print(f"{name}: {len(th)} rays, {total} ray-steps -> {path} ({os.path.getsize(path) // 1024} KiB)", flush=True)


def cfg4(pool):
    """config 4: GeoAc3D.RngDep on the 5x5x1400 grid; the rank-0 share of the 1000 az x 1000 incl fan (125 azimuths) is what one
    GPU integrates; the reference (1.1e4 steps/s per core) integrates a 2000-ray lattice of it: every 8th azimuth x every 8th inclination"""
    import tempfile
    import rngdep_data as RD
    RD.save_grid_npz(thin=1)
    grid = RD.write_grid(os.path.join(tempfile.gettempdir(), "gf"), thin=1)
    th, ph = H.fan_angles(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 124 * 0.36, phi_step=0.36)
    n_th = int(np.sum(ph == ph[0])); n_ph = len(th) // n_th
    assert n_th * n_ph == len(th)
    sel = (np.arange(0, n_ph, 8)[:, None] * n_th + np.arange(3, n_th, 8)[None, :]).ravel()
    cfgkw = dict(bounces=1, calc_amp=True, mode=0, src=(0.0, 0.0, 0.0))
    jobs = [("cfg4", H.EQ_3D_RNGDEP, tuple(grid), cfgkw, th[c], ph[c]) for c in np.array_split(sel, len(sel) // 10)]
    res = run_jobs(pool, jobs, "cfg4")
    steps = np.concatenate([r[0][0] for r in res]); flags = np.concatenate([r[0][1] for r in res]); vals = np.concatenate([r[0][2] for r in res])
    path = os.path.join(OUT, "full_cfg4.npz")
    np.savez_compressed(path, n_rays=len(th), n_theta=n_th, n_phi=n_ph, sel=sel.astype(np.int32), theta=th[sel], phi=ph[sel], bounces=1,
                        steps=steps, flags=flags, vals=vals, val_fields=np.array(VAL_FIELDS), total_steps=np.int64(steps.sum()))
    print(f"cfg4: {len(sel)} of {len(th)} rays, {int(steps.sum())} ray-steps -> {path} ({os.path.getsize(path) // 1024} KiB)", flush=True)


def cfg4_shares(pool):
    """config 4, the shares of ranks 1..7 of the 8-GPU run (geoac_amd.sharding: rank r takes the azimuths r, r + 8, ... of the 1000): of each,
    five azimuths spread over the circle x every 20th inclination = 250 rays, so that every rank's share has reference rays of its own"""
    import tempfile
    import rngdep_data as RD
    grid = RD.write_grid(os.path.join(tempfile.gettempdir(), "gf"), thin=1)
    cfgkw = dict(bounces=1, calc_amp=True, mode=0, src=(0.0, 0.0, 0.0))
    th, ph = H.fan_angles(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 999 * 0.36, phi_step=0.36)
    n_th = int(np.sum(ph == ph[0])); n_ph = len(th) // n_th
    assert n_th * n_ph == len(th) and n_ph == 999 and n_th == 1000, (n_th, n_ph)       # (repeated addition of 0.36 stops one short of 1000)
    out = {}
    for r in range(1, 8):
        az = r + 8 * np.array([3, 28, 53, 78, 103])                # azimuth indices of rank r's share (r mod 8)
        sel = (az[:, None] * n_th + np.arange(11, n_th, 20)[None, :]).ravel()
        jobs = [("cfg4", H.EQ_3D_RNGDEP, tuple(grid), cfgkw, th[c], ph[c]) for c in np.array_split(sel, len(sel) // 10)]
        res = run_jobs(pool, jobs, f"cfg4 share {r}")
        out[f"az{r}"] = az.astype(np.int32); out[f"sel{r}"] = sel.astype(np.int64); out[f"theta{r}"] = th[sel]; out[f"phi{r}"] = ph[sel]
        out[f"steps{r}"] = np.concatenate([q[0][0] for q in res]); out[f"flags{r}"] = np.concatenate([q[0][1] for q in res]); out[f"vals{r}"] = np.concatenate([q[0][2] for q in res])
    path = os.path.join(OUT, "full_cfg4_shares.npz")
    np.savez_compressed(path, bounces=1, n_theta=n_th, n_phi=n_ph, val_fields=np.array(VAL_FIELDS), **out)
    print(f"cfg4 shares 1..7: {sum(len(out[f'sel{r}']) for r in range(1, 8))} rays -> {path} ({os.path.getsize(path) // 1024} KiB)", flush=True)


def cfg4_lattice(pool):
    """config 4, the WHOLE 999 az x 1000 incl fan (what one GPU integrates at N = 1 and the eight ranks together at N = 8): a lattice of
    100 azimuths x 100 inclinations = 10 000 rays, azimuth indices 5, 15, ..., 995 (all round the circle; the odd residues mod 8 = the shares of four of the eight ranks) x inclination
    indices 7, 17, ..., 997; none of them is a ray of full_cfg4.npz or full_cfg4_shares.npz"""
    import tempfile
    import rngdep_data as RD
    grid = RD.write_grid(os.path.join(tempfile.gettempdir(), "gf"), thin=1)
    cfgkw = dict(bounces=1, calc_amp=True, mode=0, src=(0.0, 0.0, 0.0))
    th, ph = H.fan_angles(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 999 * 0.36, phi_step=0.36)
    n_th = int(np.sum(ph == ph[0])); n_ph = len(th) // n_th
    assert n_th * n_ph == len(th) and n_ph == 999 and n_th == 1000, (n_th, n_ph)
    sel = (np.arange(5, n_ph, 10)[:, None] * n_th + np.arange(7, n_th, 10)[None, :]).ravel()
    assert len(sel) == 10000
    jobs = [("cfg4", H.EQ_3D_RNGDEP, tuple(grid), cfgkw, th[c], ph[c]) for c in np.array_split(sel, len(sel) // 10)]
    res = run_jobs(pool, jobs, "cfg4 lattice")
    steps = np.concatenate([r[0][0] for r in res]); flags = np.concatenate([r[0][1] for r in res]); vals = np.concatenate([r[0][2] for r in res])
    path = os.path.join(OUT, "full_cfg4_lattice.npz")
    np.savez_compressed(path, n_rays=len(th), n_theta=n_th, n_phi=n_ph, sel=sel.astype(np.int64), theta=th[sel], phi=ph[sel], bounces=1,
                        steps=steps, flags=flags, vals=vals, val_fields=np.array(VAL_FIELDS), total_steps=np.int64(steps.sum()))
    print(f"cfg4 lattice: {len(sel)} of {len(th)} rays, {int(steps.sum())} ray-steps -> {path} ({os.path.getsize(path) // 1024} KiB)", flush=True)


SENS_EPS = 1e-12


def sens(pool, name):
    """second pass over the rays a fixture keeps values for: AMP sensitivity to a 1e-12 relative change of theta"""
    path = os.path.join(OUT, f"full_{name}.npz")
    g = dict(np.load(path))
    k = VAL_FIELDS.index("AMP")
    if name == "cfg4":
        import tempfile
        import rngdep_data as RD
        grid = tuple(RD.write_grid(os.path.join(tempfile.gettempdir(), "gf"), thin=1))
        th, ph = g["theta"], g["phi"]
        cfgkw = dict(bounces=1, calc_amp=True, mode=0, src=(0.0, 0.0, 0.0))
        chunks = np.array_split(np.arange(len(th)), len(th) // 10)
        jobs = [(name, H.EQ_3D_RNGDEP, grid, cfgkw, th[c], ph[c], SENS_EPS) for c in chunks]
    else:
        spec = FANS[name]
        tha, pha = H.fan_angles(**spec["fan"])
        vi = g["vals_idx"]
        th, ph = tha[vi], pha[vi]
        cfgkw = dict(bounces=spec["bounces"], calc_amp=True, mode=0)
        chunks = np.array_split(np.arange(len(th)), max(1, len(th) // 90))
        jobs = [(name, spec["eq"], None, cfgkw, th[c], ph[c], SENS_EPS) for c in chunks]
    res = run_jobs(pool, jobs, f"sens_{name}")
    amp2 = np.concatenate([r[0][2][..., k] for r in res])
    steps2 = np.concatenate([r[0][0] for r in res])
    amp = g["vals"][..., k]
    with np.errstate(divide="ignore", invalid="ignore"):
        sv = np.where(amp != 0.0, np.abs(amp2 - amp) / np.abs(amp), 0.0)
    ref_steps = g["steps"] if name == "cfg4" else g["steps"][g["vals_idx"]]
    sv = np.where(steps2 == ref_steps, sv, np.inf)          # a perturbation that changes a step count: knife-edge ray, no bound
    g["amp_sens"] = sv.astype(np.float32)
    g["sens_eps"] = np.float64(SENS_EPS)
    np.savez_compressed(path, **g)
    print(f"sens_{name}: {int((sv > 2.5e-7).sum())} of {int((amp != 0).sum())} arrivals move by > 2.5e-7 for a {SENS_EPS:g} change of theta; "
          f"{int(np.isinf(sv).sum())} change a step count -> {path}", flush=True)


def exempt(listfile):
    """third pass, no integration: name the arrivals whose amplitude no other arithmetic can match to 1e-6.  `listfile` (tools/amp_loose.py on
    the GPU box) lists per fixture the arrivals where the HIP path's amplitude is beyond 1e-6 of the reference's: (row of the value table, leg,
    error).  An arrival is admitted into the fixture's `amp_exempt` ONLY on the reference's own evidence - 4 x its amp_sens (running maximum
    over the ray's legs so far; `sens` pass above: the compiled reference against itself under a 1e-12 change of theta) must cover the error -
    and the list may not exceed max(3, 1e-4 x arrivals).  tests/parity.py then fails on any loose arrival that is not named here."""
    import json
    lists = json.load(open(listfile))
    for name in ("metric", "cfg2", "cfg3", "cfg4"):
        path = os.path.join(OUT, f"full_{name}.npz")
        g = dict(np.load(path))
        if "amp_sens" not in g:
            continue
        sens = np.maximum.accumulate(np.asarray(g["amp_sens"], dtype=np.float64), axis=1)
        fl = g["flags"][g["vals_idx"]] if "vals_idx" in g else g["flags"]
        n_arr = int(((fl & 1) > 0).sum())
        rows = []
        for row, leg, err, _ in lists.get(name, []):
            assert np.isfinite(sens[row, leg]) and 1e-6 < err <= 4.0 * sens[row, leg], \
                f"{name}: arrival (row {row}, leg {leg}) is off by {err:.3e} but the reference's own sensitivity only covers {4.0 * sens[row, leg]:.3e}: not an exemption, a bug"
            rows.append((row, leg))
        assert len(rows) <= max(3, 1e-4 * n_arr), f"{name}: {len(rows)} loose arrivals, more than max(3, 1e-4 x {n_arr})"
        g["amp_exempt"] = np.array(sorted(rows), dtype=np.int32).reshape(-1, 2)
        np.savez_compressed(path, **g)
        print(f"{name}: amp_exempt = {g['amp_exempt'].tolist()} ({len(rows)} of {n_arr} arrivals; cap {max(3, int(1e-4 * n_arr))}) -> {path}", flush=True)


def ring_receivers(n=64, every=8, lat0=31.0, lon0=0.0, radius_deg=2.5):
    """config 5: n receivers on a ring of 2.5 degrees of arc around the source; rank 0 of an 8-GPU run searches every 8th
    (geoac_amd.sharding.shard_receivers: round robin)"""
    az = np.arange(0, n, every) * (2.0 * np.pi / n)
    return np.stack([lat0 + radius_deg * np.cos(az), lon0 + radius_deg * np.sin(az) / np.cos(np.radians(lat0))], axis=1)


def _run_cfg5(job):
    import shutil, subprocess, tempfile
    import rngdep_data as RD
    k, lat, lon = job
    out = os.path.join(OUT, "cli", f"cfg5_{k}" if isinstance(k, str) else f"cfg5_r{k}")
    shutil.rmtree(out, ignore_errors=True); os.makedirs(out)
    args = [f"lat_src=31.0", "lon_src=0.0", f"lat_rcvr={lat!r}", f"lon_rcvr={lon!r}", "bnc_min=0", "bnc_max=2", "verbose=True"]
    with tempfile.TemporaryDirectory() as td:
        RD.write_grid_global(td)
        t0 = time.time()
        r = subprocess.run([os.path.join(H.ORACLE_DIR, "_ref", "GeoAcGlobal.RngDep"), "-eig_search", "g", "loc_lat.dat", "loc_lon.dat"] + args,
                           cwd=td, check=True, stdout=subprocess.PIPE)
        with open(os.path.join(out, "LOG.txt"), "wb") as fh:
            fh.write(r.stdout)
        for f in sorted(os.listdir(td)):
            if f.endswith(".dat") and not f.startswith("loc_"):
                shutil.copy(os.path.join(td, f), os.path.join(out, f))
    with open(os.path.join(out, "ARGS"), "w") as fh:
        fh.write("GeoAcGlobal.RngDep\n-eig_search\n" + "\n".join(args) + "\n")
    return k, time.time() - t0, sorted(os.listdir(out))


def cfg5(nproc):
    """config 5: GeoAcGlobal.RngDep -eig_search, bounces 0..2, the 8 rank-0 receivers of the 64-ring; the reference binary itself,
    one process per receiver (verbose log = the iteration sequence, _results.dat, _Eigenray-N.dat)"""
    rc = ring_receivers()
    with mp.get_context("fork").Pool(nproc) as pool:
        for k, dt, files in pool.imap_unordered(_run_cfg5, [(k, float(rc[k, 0]), float(rc[k, 1])) for k in range(len(rc))]):
            print(f"  cfg5 receiver {k}: {dt:.0f} s, {files}", flush=True)


def cfg5_ranks(nproc):
    """config 5, one receiver of each of the ranks 1..7 (rank r searches the receivers r, r + 8, ... of the 64-ring: the first of them),
    kept as cfg5_r8 .. cfg5_r14"""
    rc = ring_receivers(every=1)
    with mp.get_context("fork").Pool(nproc) as pool:
        for k, dt, files in pool.imap_unordered(_run_cfg5, [(7 + r, float(rc[r, 0]), float(rc[r, 1])) for r in range(1, 8)]):
            print(f"  cfg5 receiver {k} (ring position {k - 7}): {dt:.0f} s, {files}", flush=True)


def cfg5_rest(nproc):
    """config 5, every ring position the two modes above leave out (8 < p < 64, p not a multiple of 8): the other 49 receivers of the
    64-ring, kept as cfg5_p<ring position>.  With them every receiver of the ring is pinned to the reference binary."""
    rc = ring_receivers(every=1)
    todo = [p for p in range(8, 64) if p % 8 and not os.path.exists(os.path.join(OUT, "cli", f"cfg5_p{p}", "ARGS"))]
    with mp.get_context("fork").Pool(nproc) as pool:
        for k, dt, files in pool.imap_unordered(_run_cfg5, [(f"p{p}", float(rc[p, 0]), float(rc[p, 1])) for p in todo]):
            print(f"  cfg5 ring position {k}: {dt:.0f} s, {files}", flush=True)


def main():
    args = sys.argv[1:]
    nproc = 6
    if "-j" in args:
        i = args.index("-j"); nproc = int(args[i + 1]); del args[i:i + 2]
    which = args or ["metric", "cfg2", "cfg4", "cfg3"]
    if which[0] == "exempt":
        exempt(which[1])
        return
    for w in which:
        if w == "cfg5":
            cfg5(nproc)
            continue
        if w == "cfg5_ranks":
            cfg5_ranks(nproc)
            continue
        if w == "cfg5_rest":
            cfg5_rest(nproc)
            continue
        # a fresh pool per configuration: one equation set / atmosphere per reference process
        with mp.get_context("fork").Pool(nproc) as pool:
            if w.startswith("sens_"):
                sens(pool, w[5:])
            elif w == "cfg4":
                cfg4(pool)
            elif w == "cfg4_shares":
                cfg4_shares(pool)
            elif w == "cfg4_lattice":
                cfg4_lattice(pool)
            else:
                stratified(pool, w)


if __name__ == "__main__":
    main()
