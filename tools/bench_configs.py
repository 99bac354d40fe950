"""One GPU, every configuration of BASELINE.json (`configs`), one JSON line each: ray-steps/s of the GPU path.  The headline metric
is bench.py's; this table documents the other configurations (parity for each is in tests/).  usage: bench_configs.py [cfg ...]"""
import json, os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geoac_amd as G
import harness as H


LAST = {}


def run_fan(ctx, th, ph, reps=2):
    ctx.set_angles(th, ph)
    ctx.launch()                                   # warm-up (allocations, first-touch)
    t0 = time.perf_counter()
    ms_rk4 = 0.0; epochs = 0
    for _ in range(reps):
        ctx.launch()
        tm = ctx.timing(); ms_rk4 += tm["ms_rk4"]; epochs += tm["epochs"]
    dt = (time.perf_counter() - t0) / reps
    LAST.update(ms_rk4=ms_rk4 / reps, epochs=epochs / reps)
    try:
        LAST.update(abs_table=ctx.abs_table_info())
    except Exception:
        pass
    return ctx.total_steps(), dt


HBM_PEAK, L1_PEAK = 8000.0, 64.0 * 256 * 2.4        # GB/s: HBM3E; vector L1: 64 B/clk/CU x 256 CUs x 2.4 GHz (MI355X_MICROARCH.md)


def roofline(r, grid_gather=False):
    """per-configuration roofline object (bench.py's conventions): the dominant kernel is k_rk4; duration = HIP events around its
    launches (inside libgeoac_hip).  Stratified sets: the SURVEY 8(d) contract figure, 8 B x E per step against HBM peak (the kernels are
    FP64-latency bound).  Grid sets: the binding resource is the per-lane gather of table records through the texture /
    L1 path: 4 stages x 3 fields x 4 corners x 256 B (packed Cartesian records) = 12 288 B per ray-step, against 64 B/clk/CU."""
    steps, ms = r["ray_steps"], LAST.get("ms_rk4", 0.0)
    if ms <= 0:
        return None
    E = r.get("E", 18)
    hbm = 8.0 * E * steps / (ms * 1e-3) / 1e9
    out = {"kernel": r.get("kernel", "k_rk4"), "rk4_ms_per_pass": ms, "launches_per_pass": LAST.get("epochs"), "alg_bytes_per_step": 8 * E,
           "bound": "fp64_valu_latency", "achieved": hbm, "peak": HBM_PEAK, "unit": "GB/s", "frac": hbm / HBM_PEAK, "traffic": None}
    if grid_gather:
        g = 12288.0 * steps / (ms * 1e-3) / 1e9
        out.update({"bound": "l1_gather", "achieved": g, "peak": L1_PEAK, "frac": g / L1_PEAK, "alg_bytes_per_step": 12288,
                    "hbm_contract": {"achieved": hbm, "peak": HBM_PEAK, "frac": hbm / HBM_PEAK, "alg_bytes_per_step": 8 * E}})
    return out


def cfg1():
    ctx = G.FanContext(G.EQ_2D, device=0); ctx.load_met(H.TOYATMO)
    ctx.set_params(bounces=2, calc_amp=1, mode=1)
    th, ph = G.fan_enumerate(theta_min=1.0, theta_max=10.0, theta_step=1.0, phi_min=-90.0, phi_max=-90.0)
    steps, dt = run_fan(ctx, th, ph)
    return dict(config="cfg1 GeoAc2D -prop ToyAtmo.met theta 1..10 (10 rays, WriteRays)", rays=len(th), ray_steps=steps, seconds=dt, E=6,
                kernel="k_rk4<Eq2D<true>,true,true> (one wave: the run lasts as long as its longest ray)")


def cfg2():
    ctx = G.FanContext(G.EQ_3D, device=0); ctx.load_met(H.TOYATMO)
    ctx.set_params(bounces=2, calc_amp=1, mode=0)
    th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0)
    steps, dt = run_fan(ctx, th, ph)
    return dict(config="cfg2 GeoAc3D 360 az x 90 incl, bounces=2, CalcAmp=True", rays=len(th), ray_steps=steps, seconds=dt, E=12,
                kernel="k_rk4<Eq3DPair,true,false> || k_rk4<Eq3D<true>,true,false>")


def cfg3():
    ctx = G.FanContext(G.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO)
    ctx.set_params(bounces=3, calc_amp=1, mode=0, src=(0.0, 30.0, 0.0))
    th, ph = G.fan_enumerate(theta_min=0.25, theta_max=45.0, theta_step=0.25, phi_min=-180.0, phi_max=179.5, phi_step=0.5)
    steps, dt = run_fan(ctx, th, ph, reps=1)
    return dict(config="cfg3 GeoAcGlobal 720 az x 180 incl, bounces=3, CalcAmp=True", rays=len(th), ray_steps=steps, seconds=dt, E=18,
                kernel="k_rk4<EqGlobal<true>,true,false>")


def cfg4(thin=1):
    import rngdep_data as RD
    grid = RD.write_grid(os.path.join(tempfile.gettempdir(), f"gdb{thin}"), short_paths=False, thin=thin)
    ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0); ctx.load_grid(*grid)
    ctx.set_params(bounces=1, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
    # one GPU's share of the 1000 az x 1000 incl fan sharded over 8 GPUs: 125 azimuths
    th, ph = G.fan_enumerate(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 124 * 0.36, phi_step=0.36)
    steps, dt = run_fan(ctx, th, ph, reps=1)
    nz = 1400 // thin
    return dict(config=f"cfg4 GeoAc3D.RngDep 5x5x{nz} grid, 1/8 of 1000 az x 1000 incl (125 az), bounces=1, CalcAmp=True", rays=len(th), ray_steps=steps, seconds=dt,
                E=18, kernel="k_rk4<Eq3DRngDep<true,1,true>,false,false>", grid=True)


def cfg4_full():
    """the whole 1000 az x 1000 incl fan of config 4 on ONE GPU (999 000 rays; the path chunks take most of the device memory)"""
    import rngdep_data as RD
    grid = RD.write_grid(os.path.join(tempfile.gettempdir(), "gdb1"), short_paths=False, thin=1)
    ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0); ctx.load_grid(*grid)
    ctx.set_params(bounces=1, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
    th, ph = G.fan_enumerate(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 999 * 0.36, phi_step=0.36)
    steps, dt = run_fan(ctx, th, ph, reps=1)
    return dict(config="cfg4 GeoAc3D.RngDep 5x5x1400 grid, the whole 1000 az x 1000 incl fan on one GPU, bounces=1, CalcAmp=True", rays=len(th), ray_steps=steps, seconds=dt,
                E=18, kernel="k_rk4<Eq3DRngDep<true,1,true>,false,false>", grid=True)


def cfg4_350():
    """the same fan on the thinned 5x5x350 grid of the small parity fixtures (table 9.6 MB instead of 38 MB)"""
    return cfg4(thin=4)


def cfg5():
    import rngdep_data as RD
    grid = RD.write_grid_global(os.path.join(tempfile.gettempdir(), "ggb"), short_paths=False)
    ctx = G.FanContext(G.EQ_GLOBAL_RNGDEP, device=0); ctx.load_grid(*grid)
    ctx.set_params(src=(0.0, 31.0, 0.0))
    # one GPU's share of the 64-receiver ring sharded over 8 GPUs by receiver (geoac_amd.sharding.shard_receivers: round robin): rank 0
    az = np.arange(0, 64, 8) * (2.0 * np.pi / 64)
    rcv = np.stack([31.0 + 2.5 * np.cos(az), 2.5 * np.sin(az) / np.cos(np.radians(31.0))], axis=1)
    t0 = time.perf_counter(); out = ctx.eig_search(rcv, bnc_min=0, bnc_max=2); dt = time.perf_counter() - t0
    st = out["stats"]
    # roofline of a latency-bound search: a fan launch lasts as long as its longest ray, and a ray advances one RK4 stage per pass of ONE wave
    # through the stage loop - 1 113 instructions of the eight-lane kernel at 4 cycles each (wave64 on a 16-lane SIMD) = 4 452 cycles, i.e.
    # at most 2.4e9 / (4 x 4 452) = 1.35e5 steps/s along a ray.  achieved = the critical path's steps (sum over the launches of the longest
    # ray) / wall seconds of the whole search, host decisions and copies included.
    peak = 2.4e9 / (4 * 1113 * 4)
    ach = st["critical_steps"] / dt
    return dict(config="cfg5 GeoAcGlobal.RngDep -eig_search, 8 of 64 receivers (every 8th) on a 2.5 deg ring, bounces 0..2", rays=st["rays"], ray_steps=st["steps"],
                seconds=dt, eigenrays=int(len(out["eig"])), fan_launches=st["launches"],
                roofline={"kernel": "k_rk4<EqGlobalRngDepOct,false,false> (221 of the 295 launches; k_rk4<EqGlobalRngDep<false,4,false,true>,...> for the amplitude-less scans)",
                          "bound": "wave_issue_serial", "critical_ray_steps": st["critical_steps"], "achieved": ach, "peak": peak, "unit": "ray-steps/s along the critical path",
                          "frac": ach / peak, "stage_instructions": 1113, "traffic": None})


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"]          # (also: cfg4_350, cfg4_full)
    for w in which:
        LAST.clear()
        r = globals()[w]()
        r["ray_steps_per_s"] = r["ray_steps"] / r["seconds"]
        rf = roofline(r, grid_gather=bool(r.pop("grid", False))) if w != "cfg5" else r.get("roofline")
        r.pop("E", None); r.pop("kernel", None)
        if rf:
            r["roofline"] = rf
        if LAST.get("abs_table"):
            r["abs_table"] = LAST["abs_table"]
        print(json.dumps(r), flush=True)
