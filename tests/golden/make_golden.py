"""Generates tests/golden/*.npz from the COMPILED, UNMODIFIED reference (oracle/_ref/libref_*.so, built
by oracle/Makefile from /root/reference).  Run here only (the reference does not travel):

    make -C oracle all && python tests/golden/make_golden.py

The fixtures are data: launch angles + configuration in, full-precision arrival records / samples /
probe values out.  tests/test_oracle_golden.py pins the plain-C oracle to them bit for bit;
the GPU parity tests compare the HIP path with the same vectors.
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import harness as H  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

THETAS = [0.5, 5.0, 15.0, 25.0, 35.0, 45.0]
PHIS = [-90.0, 0.0, 37.0]


def small_fan():
    th = np.array([t for p in PHIS for t in THETAS])
    ph = np.array([p for p in PHIS for t in THETAS])
    return th, ph


def cfg_dict(cfg):
    return dict(z_grnd=cfg.z_grnd, tweak_abs=cfg.tweak_abs, freq=cfg.freq, vert_limit=cfg.vert_limit,
                range_limit=cfg.range_limit, src=np.array(list(cfg.src)), bounces=cfg.bounces,
                calc_amp=cfg.calc_amp, mode=cfg.mode)


def main():
    for eq in (H.EQ_GLOBAL, H.EQ_3D, H.EQ_2D):
        name = H.EQ_NAMES[eq]
        R = H.RefShim(eq)
        th, ph = small_fan()
        out = {"theta": th, "phi": ph}
        # fan records, both CalcAmp modes, the three output modes (Q7)
        for amp in (1, 0):
            for mode in (0, 1, 3):
                cfg = H.make_cfg(eq, calc_amp=bool(amp), mode=mode)
                steps, rec, smp, nsmp = R.fan(cfg, th, ph, smp_cap=40000)
                tag = f"amp{amp}_mode{mode}"
                out[f"rec_{tag}"] = rec
                out[f"steps_{tag}"] = np.int64(steps)
                out[f"nsmp_{tag}"] = np.int64(nsmp)
                if amp == 1 and mode == 3:
                    sel = np.arange(0, len(smp), 4)          # every 4th sample row keeps the fixture small
                    out[f"smp_idx_{tag}"] = sel
                    out[f"smp_{tag}"] = smp[sel]
        # a run with non-default ground / source / absorption / limits
        if eq == H.EQ_GLOBAL:
            src = (1.0, 41.131, -112.896)
        elif eq == H.EQ_3D:
            src = (0.0, 0.0, 1.0)
        else:
            src = (1.0, 0.0, 0.0)
        cfg = H.make_cfg(eq, bounces=1, calc_amp=True, mode=0, src=src, z_grnd=0.3, tweak_abs=0.5, freq=0.5,
                         range_limit=800.0)
        steps, rec, _, _ = R.fan(cfg, th, ph)
        out["rec_alt"] = rec
        out["steps_alt"] = np.int64(steps)
        for k, v in cfg_dict(cfg).items():
            out[f"altcfg_{k}"] = np.asarray(v)
        # stepper rows of one leg (every 50th row + the last three)
        cfg = H.make_cfg(eq, calc_amp=True)
        k, rows = R.trace_leg0(cfg, 15.0, -90.0)
        sel = np.unique(np.concatenate([np.arange(0, abs(k) + 1, 50), np.arange(abs(k) - 2, abs(k) + 1)]))
        out["trace_k"] = np.int64(k)
        out["trace_idx"] = sel
        out["trace_rows"] = rows[sel]
        # atmosphere + absorption probes
        t = R.tables()
        rng = np.random.default_rng(12345)
        x = rng.uniform(t["x"][0] - 0.5, t["x"][-1] + 0.5, 1000)
        x[:8] = [t["x"][0], t["x"][-1], t["x"][1], t["x"][700], t["x"][0] - 1, t["x"][-1] + 1, t["x"][3], t["x"][4]]
        o9, rho = R.atmo_probe(x)
        out["probe_x"] = x
        out["probe_out9"] = o9
        out["probe_rho"] = rho
        xa = rng.uniform(t["x"][0], t["x"][-1], 200)
        fa = 10.0 ** rng.uniform(-2, 1, 200)
        out["abs_x"] = xa
        out["abs_f"] = fa
        out["abs_alpha"] = R.absorption_probe(xa, fa, 0.0, 0.3)
        for key in ("x", "T", "u", "v", "rho", "sT", "su", "sv", "srho"):
            out[f"tab_{key}"] = t[key]
        path = os.path.join(OUT, f"{name}_small.npz")
        np.savez_compressed(path, **out)
        print(name, "->", path, os.path.getsize(path) // 1024, "KiB")


def main_rngdep():
    """3D.RngDep: synthetic 5x5 grid (tests/rngdep_data.py), compiled reference libref_3drd.so"""
    import tempfile
    import rngdep_data as RD
    RD.save_grid_npz()
    grid = RD.write_grid(os.path.join(tempfile.gettempdir(), "gd"))
    eq = H.EQ_3D_RNGDEP
    R = H.RefShim(eq, grid=grid)
    out = {}
    rng = np.random.default_rng(2024)
    n = 400
    x = rng.uniform(-1100, 1100, n); y = rng.uniform(-900, 900, n); z = rng.uniform(-1, 141, n)
    x[:6] = [-1000, 1000, 0, 500, -500, 250]; y[:6] = [-800, 800, 0, -400, 400, 0]; z[:6] = [0, 139.6, 10, 0.4, 70, 0.0]
    o30, a8 = R.grid_probe(x, y, z)
    out.update(probe_x=x, probe_y=y, probe_z=z, probe_out30=o30, probe_api8=a8)
    th = np.array([3.0, 9.0, 16.0, 24.0, 31.0, 40.0]); ph = np.array([-90.0, -35.0, 20.0, 75.0, 130.0, -160.0])
    out.update(theta=th, phi=ph)
    for amp in (1, 0):
        for mode in (0, 3):
            cfg = H.make_cfg(eq, bounces=1, calc_amp=bool(amp), mode=mode, src=(0.0, 0.0, 0.0))
            steps, rec, smp, nsmp = R.fan(cfg, th, ph, smp_cap=40000)
            tag = f"amp{amp}_mode{mode}"
            out[f"rec_{tag}"] = rec; out[f"steps_{tag}"] = np.int64(steps); out[f"nsmp_{tag}"] = np.int64(nsmp)
            if amp == 1 and mode == 3:
                sel = np.arange(0, len(smp), 4)
                out[f"smp_idx_{tag}"] = sel; out[f"smp_{tag}"] = smp[sel]
    # off-centre source, raised ground (enters the wind taper in this main), tighter box
    cfg = H.make_cfg(eq, bounces=2, calc_amp=True, mode=0, src=(120.0, -60.0, 1.0), freq=0.4, tweak_abs=0.6,
                     xy_limits=(-700.0, 900.0, -600.0, 700.0))
    steps, rec, _, _ = R.fan(cfg, th, ph)
    out["rec_alt"] = rec; out["steps_alt"] = np.int64(steps)
    path = os.path.join(OUT, "3drd_small.npz")
    np.savez_compressed(path, **out)
    print("3drd ->", path, os.path.getsize(path) // 1024, "KiB;", os.path.getsize(RD.GRID_NPZ) // 1024, "KiB grid")


def main_globalrd():
    """Global.RngDep: synthetic 5x5 lat/lon grid (tests/rngdep_data.py), compiled reference libref_globalrd.so"""
    import tempfile
    import rngdep_data as RD
    RD.save_grid_global_npz()
    grid = RD.write_grid_global(os.path.join(tempfile.gettempdir(), "gg"))
    eq = H.EQ_GLOBAL_RNGDEP
    R = H.RefShim(eq, grid=grid)
    out = {}
    rng = np.random.default_rng(2025)
    n = 400
    r = 6370.0 + rng.uniform(-1, 141, n); lat = np.radians(rng.uniform(24, 38, n)); lon = np.radians(rng.uniform(-9, 9, n))
    r[:6] = 6370.0 + np.array([0, 139.6, 10, 0.4, 70, 0.0]); lat[:6] = np.radians([25, 37, 31, 28, 34, 29.5]); lon[:6] = np.radians([-8, 8, 0, 4, -4, 0])
    o30, a8 = R.grid_probe(r, lat, lon)
    out.update(probe_r=r, probe_lat=lat, probe_lon=lon, probe_out30=o30, probe_api8=a8)
    th = np.array([3.0, 9.0, 16.0, 24.0, 31.0, 40.0]); ph = np.array([-90.0, -35.0, 20.0, 75.0, 130.0, -160.0])
    out.update(theta=th, phi=ph)
    for amp in (1, 0):
        for mode in (0, 3):
            cfg = H.make_cfg(eq, bounces=1, calc_amp=bool(amp), mode=mode, src=(0.0, 31.0, 0.0))
            steps, rec, smp, nsmp = R.fan(cfg, th, ph, smp_cap=40000)
            tag = f"amp{amp}_mode{mode}"
            out[f"rec_{tag}"] = rec; out[f"steps_{tag}"] = np.int64(steps); out[f"nsmp_{tag}"] = np.int64(nsmp)
            if amp == 1 and mode == 3:
                sel = np.arange(0, len(smp), 4)
                out[f"smp_idx_{tag}"] = sel; out[f"smp_{tag}"] = smp[sel]
    # off-centre elevated source, raised ground, other frequency, tighter lat/lon box (radians, as the break check compares them)
    cfg = H.make_cfg(eq, bounces=2, calc_amp=True, mode=0, src=(1.5, 29.0, 1.0), z_grnd=0.3, freq=0.4, tweak_abs=0.6,
                     xy_limits=tuple(np.radians([26.0, 36.5, -7.0, 6.0])))
    steps, rec, _, _ = R.fan(cfg, th, ph)
    out["rec_alt"] = rec; out["steps_alt"] = np.int64(steps)
    path = os.path.join(OUT, "globalrd_small.npz")
    np.savez_compressed(path, **out)
    print("globalrd ->", path, os.path.getsize(path) // 1024, "KiB;", os.path.getsize(RD.GRID_GLOBAL_NPZ) // 1024, "KiB grid")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "globalrd":
        main_globalrd()
    elif len(sys.argv) > 1 and sys.argv[1] == "rngdep":
        main_rngdep()
    else:
        main()
