"""GPU edge cases through the C ABI, checked against the plain-C oracle: single ray, ragged fan sizes around the wave width,
zero bounces, many bounces, fans in which every ray leaves the region at once, a context reused with other parameters and angle
sets, unsorted and duplicated launch angles (results come back in the caller's order), and the error paths."""
import numpy as np
import pytest

import harness as H
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
