"""GPU parity of the range-dependent spherical set (GeoAcGlobal.RngDep) against golden vectors from the compiled reference:
synthetic 5x5 lat/lon grid of perturbed profiles, 3 x 4 degree cells (tests/rngdep_data.py)."""
import numpy as np
import pytest

import harness as H
from geoac_amd.api import DEFAULT_OPTIONS as OPT      # launch-plan options of the contexts the tests create (geoac_set_option)
import rngdep_data as RD
from parity import compare_records, max_rel_errors

pytestmark = pytest.mark.gpu
EQ = H.EQ_GLOBAL_RNGDEP


@pytest.fixture(scope="module")
def gold():
    return np.load(f"{H.GOLDEN_DIR}/globalrd_small.npz")


@pytest.fixture(scope="module")
def grid(tmp_path_factory):
    return RD.write_grid_global(str(tmp_path_factory.mktemp("gg")), short_paths=False)


def _ctx(grid, **params):
    import geoac_amd as G
    ctx = G.FanContext(G.EQ_GLOBAL_RNGDEP, device=0)
    ctx.load_grid(*grid, z_grnd=0.0)          # -prop loads before any z_grnd= is seen (GeoAcGlobal.RngDep_main.cpp:133)
    ctx.set_params(**params)
    return ctx


@pytest.mark.parametrize("amp", [1, 0])
def test_globalrd_fan_vs_golden(gold, grid, amp):
    ctx = _ctx(grid, bounces=1, calc_amp=amp, mode=0, src=(0.0, 31.0, 0.0))
    rec, steps = ctx.run(gold["theta"], gold["phi"])
    want = gold[f"rec_amp{amp}_mode0"]
    E = 18 if amp else 6
    print("globalrd", amp, max_rel_errors(rec, want, E, 0))
    assert steps == int(gold[f"steps_amp{amp}_mode0"])
    compare_records(rec, want, E=E, hidx=0)


def test_globalrd_alt_config_vs_golden(gold, grid):
    ctx = _ctx(grid, bounces=2, calc_amp=1, mode=0, src=(1.5, 29.0, 1.0), z_grnd=0.3, freq=0.4, tweak_abs=0.6,
               xy_limits=tuple(np.radians([26.0, 36.5, -7.0, 6.0])))
    rec, steps = ctx.run(gold["theta"], gold["phi"])
    print("globalrd alt", max_rel_errors(rec, gold["rec_alt"], 18, 0))
    assert steps == int(gold["steps_alt"])
    compare_records(rec, gold["rec_alt"], E=18, hidx=0)


def test_globalrd_write_rays_caustics_vs_golden(gold, grid):
    ctx = _ctx(grid, bounces=1, calc_amp=1, mode=3, src=(0.0, 31.0, 0.0))
    rec, steps = ctx.run(gold["theta"], gold["phi"])
    compare_records(rec, gold["rec_amp1_mode3"], E=18, hidx=0)
    smp = ctx.fetch_samples()
    assert len(smp) == int(gold["nsmp_amp1_mode3"])
    gs = smp[gold["smp_idx_amp1_mode3"]]; ws = gold["smp_amp1_mode3"]
    assert np.array_equal(gs[:, :4], ws[:, :4])
    for col in range(4, 10):
        d = np.abs(gs[:, col] - ws[:, col])
        scale = np.maximum(np.abs(ws[:, col]), 1e-3 * max(np.abs(ws[:, col]).max(), 1e-30))
        if col == 7:
            ray_rows = ws[:, 3] == 0        # amplitude in dB on raypath rows: absolute 20 log10(1 + 1e-6); travel time on caustic rows
            assert (d[ray_rows] <= 8.7e-6 + 1e-6 * np.abs(ws[ray_rows, col])).all()
            assert (d[~ray_rows] / scale[~ray_rows] <= 1e-6).all()
        else:
            assert (d / scale).max() <= 1e-6, (col, (d / scale).max())


@pytest.mark.parametrize("lanes", [1, 2, 4, "coop", "dense", "coop+sub"])
def test_every_lanes_per_ray_variant_vs_golden(gold, grid, lanes, monkeypatch):
    if lanes == "coop+sub":
        # the launch plan of saturated fans forced on the small one: cooperative LDS-DMA gather, 512-row epochs in four sub-epochs handed from
        # workgroup to workgroup (sample and caustic events carried across the hand-off)
        monkeypatch.setitem(OPT, "SUB_MIN_WAVES", "0")
        monkeypatch.setitem(OPT, "SUB_EPOCHS", "4")
        monkeypatch.setitem(OPT, "S_ROWS", "512")
        lanes = "coop"
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
    ctx = _ctx(grid, bounces=1, calc_amp=1, mode=3, src=(0.0, 31.0, 0.0))
    rec, steps = ctx.run(gold["theta"], gold["phi"])
    assert steps == int(gold["steps_amp1_mode3"])
    compare_records(rec, gold["rec_amp1_mode3"], E=18, hidx=0)
    assert len(ctx.fetch_samples()) == int(gold["nsmp_amp1_mode3"])


def test_eight_lane_kernel_gives_the_four_lane_kernels_bits(gold, grid, monkeypatch):
    """small arrivals-only fans with amplitudes run eight lanes per ray (four cell corners x the two launch-angle systems, EqGlobalRngDepOct);
    GEOAC_OCT=0 keeps them on the four-lane kernel: same records bit for bit, and both within tolerance of the golden ones"""
    out = {}
    monkeypatch.setitem(OPT, "HEX", "0")
    for oct_on in ("1", "0"):
        monkeypatch.setitem(OPT, "OCT", oct_on)
        ctx = _ctx(grid, bounces=2, calc_amp=1, mode=0, src=(0.0, 31.0, 0.0))
        out[oct_on] = ctx.run(gold["theta"], gold["phi"])
    assert out["1"][1] == out["0"][1]
    assert np.array_equal(out["1"][0], out["0"][0])
    # ... and the sixteen-lane kernel (one field of one corner per lane, EqGlobalRngDepHex: the default for fans this small) the same bits again
    monkeypatch.setitem(OPT, "OCT", "1"); monkeypatch.setitem(OPT, "HEX", "1")
    ctx = _ctx(grid, bounces=2, calc_amp=1, mode=0, src=(0.0, 31.0, 0.0))
    rec16, steps16 = ctx.run(gold["theta"], gold["phi"])
    assert steps16 == out["1"][1]
    assert np.array_equal(rec16, out["1"][0])
    # a fan that is not a multiple of four rays (part-filled wave), and one ray alone
    for n in (5, 1):
        got = {}
        for hx in ("1", "0"):
            monkeypatch.setitem(OPT, "HEX", hx)
            ctx = _ctx(grid, bounces=2, calc_amp=1, mode=0, src=(0.0, 31.0, 0.0))
            got[hx] = ctx.run(gold["theta"][:n], gold["phi"][:n])
        assert got["1"][1] == got["0"][1] and np.array_equal(got["1"][0], got["0"][0])


def test_sixteen_lane_scan_kernel_gives_the_four_lane_kernels_bits(gold, grid, monkeypatch):
    """amplitude-less arrivals-only fans of a few hundred rays (the inclination scans of an eigenray search) split the table evaluation over sixteen lanes
    (EqGlobalRngDepScan16); HEX=0 keeps them on the four-lane kernel: the same records bit for bit, also for a part-filled wave and a single ray"""
    for n in (len(gold["theta"]), 5, 1):
        got = {}
        for hx in ("1", "0"):
            monkeypatch.setitem(OPT, "HEX", hx)
            ctx = _ctx(grid, bounces=2, calc_amp=0, mode=0, src=(0.0, 31.0, 0.0))
            got[hx] = ctx.run(gold["theta"][:n], gold["phi"][:n])
        assert got["1"][1] == got["0"][1] and np.array_equal(got["1"][0], got["0"][0])



@pytest.mark.parametrize("amp", [0, 1])
def test_leg_records_do_not_depend_on_the_number_of_bounces(grid, amp):
    """A ray launched with b bounces goes through the states of the same ray launched with fewer: its records of legs 0 .. a are, bit for bit, the
    records of the fan with a bounces (a broken ray's later legs are empty either way).  The eigenray scheduler relies on it: inclination scans
    of one round that differ only in the bounce count are integrated once, with the largest (geoac_eigenray.cpp, serve)."""
    th = np.linspace(1.0, 40.0, 79); ph = np.full_like(th, -77.0)
    recs = {}
    for b in (0, 1, 2):
        ctx = _ctx(grid, bounces=b, calc_amp=amp, mode=0, src=(0.0, 31.0, 0.0))
        recs[b] = ctx.run(th, ph)[0].copy(); ctx.close()
    for a in (0, 1):
        for b in range(a + 1, 3):
            assert np.array_equal(recs[b][:, :a + 1].view(np.uint64), recs[a].view(np.uint64)), (a, b)
