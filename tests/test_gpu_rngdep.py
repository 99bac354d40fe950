"""GPU parity of the range-dependent Cartesian set (GeoAc3D.RngDep) against golden vectors from the compiled reference:
synthetic 5x5 grid of perturbed profiles with non-square cells (tests/rngdep_data.py)."""
import numpy as np
import pytest

import harness as H
from geoac_amd.api import DEFAULT_OPTIONS as OPT      # launch-plan options of the contexts the tests create (geoac_set_option)
import rngdep_data as RD
from parity import compare_records, max_rel_errors

pytestmark = pytest.mark.gpu
EQ = H.EQ_3D_RNGDEP


@pytest.fixture(scope="module")
def gold():
    return np.load(f"{H.GOLDEN_DIR}/3drd_small.npz")


@pytest.fixture(scope="module")
def grid(tmp_path_factory):
    return RD.write_grid(str(tmp_path_factory.mktemp("gd")))


def _ctx(grid, **params):
    import geoac_amd as G
    z_grnd = params.get("z_grnd", 0.0)
    ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0)
    ctx.load_grid(*grid, z_grnd=z_grnd)
    ctx.set_params(**params)
    return ctx


@pytest.mark.parametrize("amp", [1, 0])
def test_rngdep_fan_vs_golden(gold, grid, amp):
    ctx = _ctx(grid, bounces=1, calc_amp=amp, mode=0, src=(0.0, 0.0, 0.0))
    rec, steps = ctx.run(gold["theta"], gold["phi"])
    want = gold[f"rec_amp{amp}_mode0"]
    E = 18 if amp else 6
    print("3drd", amp, max_rel_errors(rec, want, E, 2))
    assert steps == int(gold[f"steps_amp{amp}_mode0"])
    compare_records(rec, want, E=E, hidx=2)


def test_rngdep_alt_config_vs_golden(gold, grid):
    ctx = _ctx(grid, bounces=2, calc_amp=1, mode=0, src=(120.0, -60.0, 1.0), freq=0.4, tweak_abs=0.6,
               xy_limits=(-700.0, 900.0, -600.0, 700.0))
    rec, steps = ctx.run(gold["theta"], gold["phi"])
    print("3drd alt", max_rel_errors(rec, gold["rec_alt"], 18, 2))
    assert steps == int(gold["steps_alt"])
    compare_records(rec, gold["rec_alt"], E=18, hidx=2)


def test_rngdep_write_rays_caustics_vs_golden(gold, grid):
    ctx = _ctx(grid, bounces=1, calc_amp=1, mode=3, src=(0.0, 0.0, 0.0))
    rec, steps = ctx.run(gold["theta"], gold["phi"])
    compare_records(rec, gold["rec_amp1_mode3"], E=18, hidx=2)
    smp = ctx.fetch_samples()
    assert len(smp) == int(gold["nsmp_amp1_mode3"])
    gs = smp[gold["smp_idx_amp1_mode3"]]; ws = gold["smp_amp1_mode3"]
    assert np.array_equal(gs[:, :4], ws[:, :4])
    ray_rows = ws[:, 3] == 0
    for col in range(4, 10):
        d = np.abs(gs[:, col] - ws[:, col])
        scale = np.maximum(np.abs(ws[:, col]), 1e-3 * max(np.abs(ws[:, col]).max(), 1e-30))
        if col == 7:
            assert (d[ray_rows] <= 8.7e-6 + 1e-6 * np.abs(ws[ray_rows, col])).all()
            assert (d[~ray_rows] / scale[~ray_rows] <= 1e-6).all()
        else:
            assert (d / scale).max() <= 1e-6, (col, (d / scale).max())


@pytest.mark.parametrize("lanes", [1, 2, 4, "coop", "dense"])
def test_every_lanes_per_ray_variant_vs_golden(gold, grid, lanes, monkeypatch):
    """the grid kernels exist with 1, 2 and 4 lanes per ray (picked by fan size); force each on the golden fan"""
    if lanes == 2:
        import geoac_amd
        if not geoac_amd.has_ab_kernels():
            pytest.skip("the two-lane grid kernels are part of A/B builds only (make AB=1; GEOAC_LIB=<that build> runs this case)")
    if lanes in ("coop", "dense"):
        # one lane per ray without lane thinning, as a large fan runs: "coop" = wave-cooperative table gather through LDS (58 of the
        # wave's 64 lanes are helpers without a ray here), "dense" = the same launch with per-lane gathers
        monkeypatch.setitem(OPT, "GRID_LANES", "1")
        monkeypatch.setitem(OPT, "SPREAD", "1")
        monkeypatch.setitem(OPT, "GRID_COOP", "1" if lanes == "coop" else "0")
    else:
        monkeypatch.setitem(OPT, "GRID_LANES", str(lanes))
    ctx = _ctx(grid, bounces=1, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
    rec, steps = ctx.run(gold["theta"], gold["phi"])
    assert steps == int(gold["steps_amp1_mode0"])
    compare_records(rec, gold["rec_amp1_mode0"], E=18, hidx=2)
    ctx = _ctx(grid, bounces=1, calc_amp=0, mode=0, src=(0.0, 0.0, 0.0))
    rec, steps = ctx.run(gold["theta"], gold["phi"])
    assert steps == int(gold["steps_amp0_mode0"])
    compare_records(rec, gold["rec_amp0_mode0"], E=6, hidx=2)


def test_eight_lane_kernel_gives_the_four_lane_kernels_bits(gold, grid, monkeypatch):
    """small arrivals-only fans with amplitudes run eight lanes per ray (four cell corners x the two launch-angle systems, Eq3DRngDepOct; row k - 2
    of the quadratic intercept split between the systems' first lanes); GEOAC_OCT=0 keeps them on the four-lane kernel: same records bit for bit"""
    out = {}
    for oct_on in ("1", "0"):
        monkeypatch.setitem(OPT, "OCT", oct_on)
        ctx = _ctx(grid, bounces=2, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
        out[oct_on] = ctx.run(gold["theta"], gold["phi"])
    assert out["1"][1] == out["0"][1]
    assert np.array_equal(out["1"][0], out["0"][0])


def test_tiled_slot_order_gives_the_inclination_orders_bits(grid, monkeypatch):
    """the grid sets integrate a fan in Z-order over (inclination rank, azimuth rank) - 8 x 8 tiles of the fan per wave, fewer distinct
    (segment, cell) records per gather - and hand the records back in the caller's order; TILE=0 keeps the inclination order, SORT=0 the caller's:
    the same records bit for bit (a fan with a ragged edge: 37 inclinations x 11 azimuths, cooperative kernel forced by GRID_LANES=1)"""
    th = np.repeat(np.linspace(2.0, 38.0, 37)[None, :], 11, axis=0).reshape(-1)
    ph = np.repeat((-120.0 + 7.5 * np.arange(11))[:, None], 37, axis=1).reshape(-1)
    out = {}
    for name, opts in (("tile", {}), ("incl", {"TILE": "0"}), ("caller", {"SORT": "0"})):
        for k in ("TILE", "SORT"):
            monkeypatch.delitem(OPT, k, raising=False)
        for k, v in opts.items():
            monkeypatch.setitem(OPT, k, v)
        monkeypatch.setitem(OPT, "GRID_LANES", "1")
        ctx = _ctx(grid, bounces=1, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
        out[name] = ctx.run(th, ph)
    assert out["tile"][1] == out["incl"][1] == out["caller"][1]
    assert np.array_equal(out["tile"][0], out["incl"][0]) and np.array_equal(out["tile"][0], out["caller"][0])
