"""GPU edge cases through the C ABI, checked against the plain-C oracle: single ray, ragged fan sizes around the wave width,
zero bounces, many bounces, fans in which every ray leaves the region at once, a context reused with other parameters and angle
sets, unsorted and duplicated launch angles (results come back in the caller's order), and the error paths."""
import numpy as np
import pytest

import harness as H
from geoac_amd.api import DEFAULT_OPTIONS as OPT      # launch-plan options of the contexts the tests create (geoac_set_option)
from parity import compare_records

pytestmark = pytest.mark.gpu


def _gpu(eq, **params):
    import geoac_amd as G
    ctx = G.FanContext(eq, device=0)
    ctx.load_met(H.TOYATMO)
    ctx.set_params(**params)
    return ctx


def _check(eq, theta, phi, E, **kw):
    ctx = _gpu(eq, **kw)
    rec, steps = ctx.run(theta, phi)
    cfg = H.make_cfg(eq, bounces=kw.get("bounces", 2), calc_amp=bool(kw.get("calc_amp", 1)), mode=kw.get("mode", 0),
                     **{k: v for k, v in kw.items() if k in ("vert_limit", "range_limit", "z_grnd", "src")})
    so, ro, _, _ = H.Oracle(eq).fan(cfg, theta, phi)
    assert steps == so
    compare_records(rec, ro, E=E)
    return rec


@pytest.mark.parametrize("n", [1, 63, 64, 65, 129])
def test_ragged_fan_sizes(n):
    th = np.linspace(4.0, 40.0, n) if n > 1 else np.array([17.0])
    ph = np.full(n, -90.0) + (np.arange(n) % 7) * 11.0
    _check(H.EQ_GLOBAL, th, ph, 18, bounces=0, calc_amp=1)


def test_zero_and_many_bounces():
    th = np.array([12.0, 30.0]); ph = np.array([-90.0, 45.0])
    _check(H.EQ_3D, th, ph, 12, bounces=0, calc_amp=1)
    rec = _check(H.EQ_3D, th, ph, 4, bounces=9, calc_amp=0, range_limit=1.0e6)
    assert rec[:, :, H.REC["VALID"]].sum() >= 10        # the 30 degree ray keeps bouncing until the last leg


def test_every_ray_breaks_at_once():
    """vert_limit below the source: the first step of every ray leaves the region (no arrival rows at all)"""
    th = np.array([5.0, 20.0, 35.0]); ph = np.array([0.0, 90.0, 180.0])
    rec = _check(H.EQ_2D, th, ph, 6, bounces=2, calc_amp=1, mode=1, src=(2.0, 0.0, 0.0), vert_limit=1.0)
    assert rec[:, :, H.REC["VALID"]].sum() == 0 and (rec[:, 0, H.REC["BROKE"]] == 1).all()
    assert (rec[:, 0, H.REC["STEPS"]] == 1).all()


def test_context_reuse_and_caller_order():
    import geoac_amd as G
    ctx = _gpu(H.EQ_GLOBAL, bounces=1, calc_amp=1)
    th = np.array([33.0, 2.5, 33.0, 18.0, 9.0]); ph = np.array([10.0, -90.0, 10.0, 140.0, -20.0])       # unsorted, one duplicate
    rec1, _ = ctx.run(th, ph)
    assert np.array_equal(rec1[0], rec1[2])                           # duplicates integrate identically, wherever they sit in a wave
    so, ro, _, _ = H.Oracle(H.EQ_GLOBAL).fan(H.make_cfg(H.EQ_GLOBAL, bounces=1, calc_amp=True), th, ph)
    compare_records(rec1, ro, E=18)
    ctx.set_params(bounces=2, calc_amp=0, freq=0.5)                   # same context, other parameters and another angle set
    th2 = np.array([7.0, 41.0]); ph2 = np.array([-170.0, 60.0])
    rec2, s2 = ctx.run(th2, ph2)
    so2, ro2, _, _ = H.Oracle(H.EQ_GLOBAL).fan(H.make_cfg(H.EQ_GLOBAL, bounces=2, calc_amp=False, freq=0.5), th2, ph2)
    assert s2 == so2
    compare_records(rec2, ro2, E=6)
    ctx.set_params(bounces=1, calc_amp=1, freq=0.1)
    rec3, _ = ctx.run(th, ph)
    assert np.array_equal(rec3, rec1)                                 # and back: bit-identical to the first launch


def test_ray_order_of_the_sort_does_not_reach_the_records():
    """geoac_fan_set_angles orders the rays by inclination (a radix sort on the doubles' integer image; a comparison sort when a -0.0 or a NaN is among them) and the
    records come back in the caller's order: a few thousand rays in random order with many equal inclinations against the same fan in sorted order, and an
    inclination of -0.0 (the comparison-sort path) against +0.0"""
    rng = np.random.default_rng(5)
    th = np.round(rng.uniform(1.0, 40.0, 3000), 0); ph = np.round(rng.uniform(-180.0, 180.0, 3000), 1)
    order = np.lexsort((ph, th))
    ctx = _gpu(H.EQ_GLOBAL, bounces=1, calc_amp=1)
    rec_sorted, s1 = ctx.run(th[order], ph[order])
    rec_sorted = rec_sorted.copy()
    rec_any, s2 = ctx.run(th, ph)
    assert s1 == s2
    assert np.array_equal(rec_any[order].view(np.uint64), rec_sorted.view(np.uint64))
    a, _ = ctx.run(np.array([5.0, -0.0, 12.0, 5.0]), np.array([10.0, 20.0, 30.0, 10.0])); a = a.copy()
    b, _ = ctx.run(np.array([5.0, 0.0, 12.0, 5.0]), np.array([10.0, 20.0, 30.0, 10.0]))
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_error_paths():
    import geoac_amd as G
    ctx = G.FanContext(G.EQ_GLOBAL, device=0)
    with pytest.raises(G.GeoAcError):
        ctx.run(np.array([10.0]), np.array([0.0]))                    # no atmosphere uploaded
    ctx.load_met(H.TOYATMO)
    with pytest.raises(G.GeoAcError):
        ctx.set_params(bounces=64)                                    # more legs than the record table is laid out for
    with pytest.raises(G.GeoAcError):
        ctx.set_params(ds_min=0.0)
    fresh = G.FanContext(G.EQ_GLOBAL, device=0)
    fresh.load_met(H.TOYATMO)
    with pytest.raises(G.GeoAcError):
        fresh.launch()                                                # no launch angles yet
    ctx.set_params(bounces=0, ds_min=0.001)
    rec, steps = ctx.run(np.array([10.0]), np.array([0.0]))
    assert steps > 0 and rec.shape == (1, 1, 32)
    grid_ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0)
    with pytest.raises(G.GeoAcError):
        grid_ctx.load_met(H.TOYATMO)                                  # a 1-D profile is not an atmosphere for a grid set


def test_step_limit_exhaustion_is_an_ordinary_leg_end():
    """GeoAc_Propagate_RK4 stops after step_limit - 1 steps (GeoAc.Solver.cpp:14,25) and returns step_limit with check = false: the mains write the
    leg as an arrival.  With ray_limit = 10 (step_limit = 1000, at most 50 km of path) no ray reaches the ground or the top: every leg's STEPS is the reference's 1000,
    the launch succeeds, and the condition is reported through geoac_fan_status.  (The arrival row itself is not compared: the reference reads
    the row solution[step_limit] it never wrote; here the leg ends on the last integrated row.)"""
    import geoac_amd as G
    th = np.array([3.0, 17.0, 40.0]); ph = np.array([-90.0, 20.0, 135.0])
    ctx = _gpu(H.EQ_GLOBAL, bounces=0, calc_amp=1, ray_limit=10.0)
    rec, steps = ctx.run(th, ph)
    assert ctx.fan_status() & 1                                       # GEOAC_FAN_STEP_LIMIT
    O = H.Oracle(H.EQ_GLOBAL)
    O.set_ray_limit(10.0)
    so, ro, _, _ = O.fan(H.make_cfg(H.EQ_GLOBAL, bounces=0, calc_amp=True), th, ph)
    assert steps == so == 3 * 1000
    for f in ("STEPS", "VALID", "BROKE"):
        assert np.array_equal(rec[:, :, G.REC[f]], ro[:, :, H.REC[f]]), f
    assert (rec[:, 0, G.REC["STEPS"]] == 1000).all() and (rec[:, 0, G.REC["VALID"]] == 1).all()
    np.testing.assert_allclose(rec[:, 0, G.REC["TURN"]], ro[:, 0, H.REC["TURN"]], rtol=1e-9)
    # an ordinary run on the same context afterwards: the flag is per launch
    ctx.set_params(ray_limit=10000.0)
    rec2, _ = ctx.run(th, ph)
    assert ctx.fan_status() == 0 and (rec2[:, 0, G.REC["STEPS"]] > 1000).all()


def test_event_list_overflow_is_reported(monkeypatch):
    """the per-epoch list of raypath-sample / caustic rows of a ray holds s_rows / stride + a slack for the caustics; with the slack taken away
    (GEOAC_EV_SLACK=0) a ray with caustics overflows it: the launch must fail with GEOAC_E_CAPACITY, not return a truncated table"""
    import geoac_amd as G
    monkeypatch.setitem(OPT, "EV_SLACK", "0")
    monkeypatch.setitem(OPT, "S_ROWS", "64")
    ctx = _gpu(H.EQ_GLOBAL, bounces=1, calc_amp=1, mode=1 | 2)     # GEOAC_MODE_WRITE_RAYS | GEOAC_MODE_WRITE_CAUSTICS
    with pytest.raises(G.GeoAcError, match="event list overflowed|capacity"):
        ctx.run(np.array([5.0, 12.0, 25.0]), np.array([-90.0, -90.0, -90.0]))


def test_options_go_through_the_abi_not_the_environment(monkeypatch):
    """launch-plan options reach a context through geoac_set_option (api.FanContext(options=...), api.options()); the library ignores GEOAC_*
    environment variables unless GEOAC_DEBUG_ENV=1; an unknown key is an error; records do not depend on the option"""
    import geoac_amd as G
    assert {"S_ROWS", "COMPACT", "PAIR_FRAC", "GRID_LANES", "ABS_TABLE"} <= set(G.option_names())
    th = np.array([4.0, 21.0]); ph = np.array([-90.0, 33.0])

    def epochs(**kw):
        ctx = G.FanContext(H.EQ_GLOBAL, device=0, **kw)
        ctx.load_met(H.TOYATMO)
        ctx.set_params(bounces=0, calc_amp=1)
        rec, _ = ctx.run(th, ph)
        return rec, ctx.timing()["epochs"]
    monkeypatch.delenv("GEOAC_DEBUG_ENV", raising=False)
    monkeypatch.setenv("GEOAC_S_ROWS", "64")
    rec0, e0 = epochs()                                               # the variable is NOT read
    rec1, e1 = epochs(options={"S_ROWS": 64})
    with G.options(S_ROWS=64):
        rec2, e2 = epochs()
    assert e0 < 20 < e1 == e2
    assert np.array_equal(rec0, rec1) and np.array_equal(rec0, rec2)
    monkeypatch.setenv("GEOAC_DEBUG_ENV", "1")
    rec3, e3 = epochs()                                               # debug switch: now it is
    assert e3 == e1 and np.array_equal(rec0, rec3)
    with pytest.raises(G.GeoAcError):
        G.FanContext(H.EQ_GLOBAL, device=0, options={"NO_SUCH_KNOB": 1})
    # a value that does not parse or lies outside the knob's range is an error too - never a silent 0 (SORT=abc used to switch sorting off)
    for bad in ({"SORT": "abc"}, {"PP_BLOCKS": -1}, {"HYBRID_ROWS": 1.5}, {"PAIR_FRAC": -0.1}, {"PPFIX_CAP": 0}, {"S_ROWS": "12x"}, {"GRID_LANES": 3},
                {"SUB_EPOCHS": 0}, {"GRID_BUILD": "gpu"}, {"ABS_TABLE_TOL": "0"}, {"COMPACT": 2}):
        with pytest.raises(G.GeoAcError, match="expected"):
            G.FanContext(H.EQ_GLOBAL, device=0, options=bad)


def test_a_clone_refuses_to_launch_after_its_source_changed_its_atmosphere():
    """geoac_clone shares the source's atmosphere tables as views; once the source uploads another atmosphere (or is destroyed) the views show other data or freed
    memory - the clone's next launch must fail loudly instead"""
    import geoac_amd as G
    th = np.array([4.0, 21.0]); ph = np.array([-90.0, 33.0])
    src = G.FanContext(H.EQ_GLOBAL, device=0); src.load_met(H.TOYATMO); src.set_params(bounces=0, calc_amp=1)
    want, _ = src.run(th, ph)
    c = src.clone()
    got, _ = c.run(th, ph)
    assert np.array_equal(got, want)                                  # the clone integrates the same fan on the shared tables
    src.load_met(H.TOYATMO)                                           # a new upload (same file: the tables are rebuilt all the same)
    with pytest.raises(G.GeoAcError, match="clone"):
        c.run(th, ph)
    c2 = src.clone()
    assert np.array_equal(c2.run(th, ph)[0], want)
    src.close()
    with pytest.raises(G.GeoAcError, match="clone"):
        c2.run(th, ph)
    c.close(); c2.close()
