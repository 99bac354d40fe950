"""BASELINE.json's configurations at FULL size against fixtures made by the compiled reference (tests/golden/make_golden_full.py,
oracle/_ref/libref_*.so run here, 6 processes): every (ray, leg) of the GPU fan must reproduce the reference's
GeoAc_Propagate_RK4 step count and its VALID / BROKE outcome exactly, and travel time, attenuation, turning height, arrival
inclination, back azimuth, amplitude and range within 1e-6 relative (tests/parity.py).

  metric   GeoAcGlobal 360 az x 90 incl, bounces=2, CalcAmp=True: all 32 400 rays x 3 legs, values for all
  cfg2     GeoAc3D, same fan: all rays, values for all
  cfg3     GeoAcGlobal 720 az x 180 incl, bounces=3: counts for all 129 600 rays x 4 legs, values for every 4th azimuth
  cfg4     GeoAc3D.RngDep, 5x5x1400 grid, the rank-0 share (125 azimuths x 1000 inclinations = 124 000 rays) of the 1000 x 1000 fan:
           the GPU integrates the whole share; the reference's lattice of 2000 of its rays (every 8th azimuth x every 8th inclination)
           + 250 rays of each of the other seven ranks' shares (five of the rank's azimuths x every 20th inclination)
           + a 10 000-ray lattice of the whole 999 x 1000 fan (100 azimuths x 100 inclinations)
  cfg5     GeoAcGlobal.RngDep -eig_search, all 64 receivers of the ring: tests/test_gpu_eig_ring.py
"""
import os

import numpy as np
import pytest

import harness as H
from parity import compare_compact

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import geoac_amd
    geoac_amd.load_library()
    return geoac_amd


def _gold(name):
    return np.load(os.path.join(H.GOLDEN_DIR, f"full_{name}.npz"))


def _fan_kwargs(g):
    return {str(k): float(v) for k, v in g["fan"]}


@pytest.mark.parametrize("name,eqname", [("metric", "EQ_GLOBAL"), ("cfg2", "EQ_3D"), ("cfg3", "EQ_GLOBAL")])
def test_stratified_fan_every_ray_vs_reference(G, name, eqname):
    g = _gold(name)
    th, ph = G.fan_enumerate(**_fan_kwargs(g))
    assert len(th) == int(g["n_rays"])
    ctx = G.FanContext(getattr(G, eqname), device=0)
    ctx.load_met(H.TOYATMO)
    ctx.set_params(bounces=int(g["bounces"]), calc_amp=1, mode=0)
    rec, steps = ctx.run(th, ph)
    assert steps == int(g["total_steps"])                 # the reference's own sum of GeoAc_Propagate_RK4 return values
    err = compare_compact(rec, g)
    print(name, "ray-steps", steps, "max rel err", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in err.items()})
    if name == "metric":
        assert steps == 874273730


def test_config4_share_on_5x5x1400_grid_vs_reference(G, tmp_path):
    import rngdep_data as RD
    g = _gold("cfg4")
    grid = RD.write_grid(str(tmp_path), short_paths=False, thin=1)
    ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0)
    ctx.load_grid(*grid)
    ctx.set_params(bounces=int(g["bounces"]), calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
    th, ph = G.fan_enumerate(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 124 * 0.36, phi_step=0.36)
    assert len(th) == int(g["n_rays"]) == 124000
    sel = g["sel"]
    assert np.array_equal(th[sel], g["theta"]) and np.array_equal(ph[sel], g["phi"])
    rec, steps = ctx.run(th, ph)
    assert int(rec[..., H.REC["STEPS"]].sum()) == steps
    gold = {k: g[k] for k in ("steps", "flags", "vals", "val_fields")}
    err = compare_compact(rec, gold, idx=sel)
    print("cfg4 share:", steps, "ray-steps;", len(sel), "rays vs reference, max rel err", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in err.items()})


def test_config4_other_ranks_shares_vs_reference(G, tmp_path):
    """the shares of ranks 1..7 of the 8-GPU run (geoac_amd.sharding: rank r integrates the azimuths r, r + 8, ... of the 999): five azimuths of
    every rank's share x all 1000 inclinations on the GPU (35 000 rays: the cooperative one-lane-per-ray kernel of the full fan), 250 of each
    rank's rays against the compiled reference (tests/golden/full_cfg4_shares.npz, make_golden_full.py cfg4_shares)"""
    import rngdep_data as RD
    g = np.load(os.path.join(H.GOLDEN_DIR, "full_cfg4_shares.npz"))
    grid = RD.write_grid(str(tmp_path), short_paths=False, thin=1)
    ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0)
    ctx.load_grid(*grid)
    ctx.set_params(bounces=int(g["bounces"]), calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
    th, ph = G.fan_enumerate(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 999 * 0.36, phi_step=0.36)
    n_th = int(g["n_theta"])
    assert len(th) == n_th * int(g["n_phi"])
    az = np.concatenate([g[f"az{r}"] for r in range(1, 8)])
    assert all((g[f"az{r}"] % 8 == r).all() for r in range(1, 8))            # each rank's own azimuths
    rays = (az[:, None] * n_th + np.arange(n_th)[None, :]).ravel()
    rec, steps = ctx.run(th[rays], ph[rays])
    assert int(rec[..., H.REC["STEPS"]].sum()) == steps
    pos = {int(a): i for i, a in enumerate(az)}
    worst = {}
    for r in range(1, 8):
        sel = g[f"sel{r}"]
        assert np.array_equal(th[sel], g[f"theta{r}"]) and np.array_equal(ph[sel], g[f"phi{r}"])
        local = np.array([pos[int(q // n_th)] * n_th + int(q % n_th) for q in sel])
        gold = {"steps": g[f"steps{r}"], "flags": g[f"flags{r}"], "vals": g[f"vals{r}"], "val_fields": g["val_fields"]}
        err = compare_compact(rec, gold, idx=local)
        for k, v in err.items():
            if not isinstance(v, list):
                worst[k] = max(worst.get(k, 0.0), v)
    print("cfg4 shares of ranks 1..7:", steps, "ray-steps on the GPU;", 7 * 250, "rays vs reference, max rel err", {k: f"{v:.2e}" for k, v in worst.items()})


def test_config4_whole_fan_lattice_vs_reference(G, tmp_path):
    """a lattice over the WHOLE 999 x 1000 fan of config 4 (tests/golden/full_cfg4_lattice.npz, make_golden_full.py cfg4_lattice: azimuth indices 5, 15, ..., 995 -
    spread over the whole circle, in the shares of ranks 1, 3, 5, 7 of the 8-GPU run - x inclination indices 7, 17, ..., 997 = 10 000 rays integrated by the compiled reference):
    the GPU integrates those 100 azimuths with all their 1000 inclinations (100 000 rays: the cooperative one-lane kernel of the full fan) and every lattice ray
    must match - counts exact, values to 1e-6.  bench.py checks the same fixture against the 999 000-ray fan itself."""
    import rngdep_data as RD
    g = np.load(os.path.join(H.GOLDEN_DIR, "full_cfg4_lattice.npz"))
    grid = RD.write_grid(str(tmp_path), short_paths=False, thin=1)
    ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0)
    ctx.load_grid(*grid)
    ctx.set_params(bounces=int(g["bounces"]), calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
    th, ph = G.fan_enumerate(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 999 * 0.36, phi_step=0.36)
    n_th = int(g["n_theta"])
    assert len(th) == n_th * int(g["n_phi"]) == int(g["n_rays"])
    sel = g["sel"]
    assert len(sel) == 10000 and np.array_equal(th[sel], g["theta"]) and np.array_equal(ph[sel], g["phi"])
    az = np.unique(sel // n_th)
    assert len(az) == 100 and set(int(a) % 8 for a in az) == {1, 3, 5, 7}        # (the shares of four of the eight ranks; full_cfg4 / full_cfg4_shares hold rays of all eight)
    rays = (az[:, None] * n_th + np.arange(n_th)[None, :]).ravel()
    rec, steps = ctx.run(th[rays], ph[rays])
    assert int(rec[..., H.REC["STEPS"]].sum()) == steps
    pos = {int(a): i for i, a in enumerate(az)}
    local = np.array([pos[int(q // n_th)] * n_th + int(q % n_th) for q in sel])
    err = compare_compact(rec, {k: g[k] for k in ("steps", "flags", "vals", "val_fields")}, idx=local)
    print("cfg4 lattice of the whole fan:", steps, "ray-steps on the GPU;", len(sel), "rays vs reference, max rel err", {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in err.items()})


def test_config4_share_is_schedule_independent(G, tmp_path):
    """the config-4 share under the launch plans of the grid kernels - default (LDS-DMA cooperative gather, four sub-epochs per epoch),
    without sub-epochs, 4096-row epochs in eight sub-epochs, no compaction, per-lane gathers - gives bit-identical records"""
    import os
    import rngdep_data as RD
    grid = RD.write_grid(str(tmp_path), short_paths=False, thin=1)
    th, ph = G.fan_enumerate(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 124 * 0.36, phi_step=0.36)

    def run(env):
        with G.options(**env):                                # launch-plan options of the contexts created inside (geoac_set_option)
            ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0)      # the knobs are read when the context is created
            ctx.load_grid(*grid)
            ctx.set_params(bounces=1, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
            rec, steps = ctx.run(th, ph)
            ctx.close()
        return rec, steps
    ref, steps = run({})
    for env in ({"GEOAC_SUB_EPOCHS": "1"}, {"GEOAC_SUB_EPOCHS": "8", "GEOAC_S_ROWS": "4096"}, {"GEOAC_COMPACT": "0"}, {"GEOAC_GRID_COOP": "0"}):
        rec, st = run(env)
        assert st == steps, env
        assert np.array_equal(rec, ref), env
