"""GPU test of the device-side table builder of the range-dependent sets (geoac_gridbuild.hip) against the host
restatement of the same construction (geoac_grid_table_eq in geoac_host.cpp, itself pinned to the compiled reference's scalar API by
tests/test_host_grid.py): the natural-spline slopes are the same operations in the same order (FMA contraction off) and must agree to
the bit; the expanded cubic coefficients (long double on the host, double-double on the device) to a few units in the last place
relative to the size of the terms they are built from."""
import ctypes

import numpy as np
import pytest

import geoac_amd as G
import harness as H

pytestmark = pytest.mark.gpu
_dp = ctypes.POINTER(ctypes.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


def _synthetic_grid(nx, ny, nz, spherical, seed=7):
    rng = np.random.default_rng(seed)
    if spherical:
        x = np.radians(25.0 + np.cumsum(0.3 + 0.4 * rng.random(nx)))          # ragged node spacing on every axis
        y = np.radians(-5.0 + np.cumsum(0.3 + 0.4 * rng.random(ny)))
        z = 6370.0 + np.concatenate([[0.0], np.cumsum(0.05 + 0.2 * rng.random(nz - 1))])
    else:
        x = -400.0 + np.cumsum(20.0 + 30.0 * rng.random(nx))
        y = -300.0 + np.cumsum(20.0 + 30.0 * rng.random(ny))
        z = np.concatenate([[0.0], np.cumsum(0.05 + 0.2 * rng.random(nz - 1))])
    zz = (z - z[0])[None, None, :]
    xx = ((x - x[0]) / (x[-1] - x[0]))[:, None, None]
    yy = ((y - y[0]) / (y[-1] - y[0]))[None, :, None]
    T = 288.0 - 6.5 * np.minimum(zz, 11.0) + 15.0 * np.sin(0.21 * zz + 2.0 * xx) * np.cos(1.3 * yy) + 0.5 * rng.standard_normal((nx, ny, nz))
    u = 0.03 * np.sin(0.13 * zz + 1.7 * yy) * (1 + xx) + 1e-3 * rng.standard_normal((nx, ny, nz))
    v = 0.02 * np.cos(0.17 * zz - 2.3 * xx) * (1 + yy) + 1e-3 * rng.standard_normal((nx, ny, nz))
    rho = 1.2e-3 * np.exp(-zz / 7.0) * (1 + 0.05 * xx * yy) + 0.0 * xx
    return [np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, z, T, u, v, rho + 0.0 * T)]


@pytest.mark.parametrize("eq,shape", [(G.EQ_3D_RNGDEP, (5, 5, 60)), (G.EQ_3D_RNGDEP, (13, 9, 257)), (G.EQ_GLOBAL_RNGDEP, (7, 11, 130)),
                                      (G.EQ_GLOBAL_RNGDEP, (2, 2, 3)), (G.EQ_3D_RNGDEP, (2, 3, 1400))])
def test_device_table_matches_host_builder(eq, shape):
    nx, ny, nz = shape
    x, y, z, T, u, v, rho = _synthetic_grid(nx, ny, nz, eq == G.EQ_GLOBAL_RNGDEP)
    lib = G.load_library()
    lib.geoac_grid_table_size.restype = ctypes.c_size_t
    n = lib.geoac_grid_table_size(nx, ny, nz)
    host = np.zeros(n)
    lib.geoac_grid_table_eq.argtypes = None
    assert lib.geoac_grid_table_eq(eq, nx, ny, nz, _p(x), _p(y), _p(z), _p(T), _p(u), _p(v), _p(rho), _p(host)) == 0
    ctx = G.FanContext(eq, device=0)
    ctx.upload_atmo_3d(x, y, z, T, u, v, rho)
    dev = ctx.grid_table()
    ctx.close()
    assert dev.shape == host.shape
    nseg, nn = nz - 1, nx * ny
    main_h, main_d = host[:3 * nseg * nn * 40].reshape(3, nseg, nn, 10, 4), dev[:3 * nseg * nn * 40].reshape(3, nseg, nn, 10, 4)
    rho_h, rho_d = host[3 * nseg * nn * 40:].reshape(nseg, nn, 4, 4), dev[3 * nseg * nn * 40:].reshape(nseg, nn, 4, 4)
    # c0 (node values) and c1 (slopes) of the un-differenced cubics: bit-identical
    for a, b in ((main_h[..., 0, :2], main_d[..., 0, :2]), (rho_h[..., 0, :2], rho_d[..., 0, :2])):
        assert np.array_equal(a, b), "values / natural-spline slopes differ from the host builder"
    if eq == G.EQ_3D_RNGDEP:                                # spherical sets fold the Q12b offset into c1 of the Vx / Vy rows
        assert np.array_equal(main_h[..., 4, :2], main_d[..., 4, :2]) and np.array_equal(main_h[..., 7, :2], main_d[..., 7, :2])
    # every coefficient: the differenced cubics (Dx, Dy, Dxy) subtract neighbouring columns, so a last-place difference of a base
    # coefficient shows up divided by the node spacing: hold each to a few ulps of its BASE cubic's scale times that factor
    IX, IY = 1.0 / np.diff(x).min(), 1.0 / np.diff(y).min()
    unit = np.array([1.0, IX, IY, IX * IY, 1.0, IX, IX * IY, 1.0, IY, IX * IY])
    base = np.array([0, 0, 0, 0, 4, 4, 4, 7, 7, 7])
    for c in range(4):
        bs = np.array([np.abs(main_h[..., b, c]).max() for b in base]) + 1e-300
        err = np.abs(main_h[..., c] - main_d[..., c]).max(axis=(0, 1, 2)) / (bs * unit)
        assert err.max() < 1e-14, f"coefficient c{c}: {err}"
        bs = np.abs(rho_h[..., 0, c]).max() + 1e-300
        err = np.abs(rho_h[..., c] - rho_d[..., c]).max(axis=(0, 1)) / (bs * unit[:4])
        assert err.max() < 1e-14, f"rho coefficient c{c}: {err}"


@pytest.mark.parametrize("eq,shape", [(G.EQ_3D_RNGDEP, (7, 4, 61)), (G.EQ_GLOBAL_RNGDEP, (6, 9, 45)), (G.EQ_3D_RNGDEP, (2, 2, 3)), (G.EQ_GLOBAL_RNGDEP, (13, 3, 257))])
def test_interpolant_gathers_agree_on_ragged_grids(eq, shape):
    """the per-lane evaluator and the wave-cooperative LDS-DMA one (packed 256-byte records for the Cartesian set, 320-byte records in five
    rounds for the spherical one) on grids with ragged node spacings and other shapes than the 5 x 5 fixtures (down to the smallest legal one,
    2 x 2 x 3): bit-identical at 2048 random points, everything finite"""
    nx, ny, nz = shape
    spherical = eq == G.EQ_GLOBAL_RNGDEP
    x, y, z, T, u, v, rho = _synthetic_grid(nx, ny, nz, spherical, seed=11)
    ctx = G.FanContext(eq, device=0)
    ctx.upload_atmo_3d(x, y, z, T, u, v, rho)
    src = (0.0, float(np.degrees(x[nx // 2])), float(np.degrees(y[ny // 2]))) if spherical else (0.0, float(x[nx // 2]), float(y[ny // 2]))
    ctx.set_params(bounces=0, calc_amp=1, mode=0, src=src if spherical else (float(x[nx // 2]), float(y[ny // 2]), 0.0))
    ctx.run(np.array([30.0]), np.array([-90.0]))                 # (the probes use the launch parameters of the last fan)
    rng = np.random.default_rng(3)
    n = 2048
    a = (rng.uniform(x[0], x[-1], n), rng.uniform(y[0], y[-1], n), rng.uniform(z[0], z[-1], n))
    o0, a0 = ctx.probe_grid(*a, coop=False)
    o1, a1 = ctx.probe_grid(*a, coop=True)
    assert np.isfinite(o0).all() and np.isfinite(a0).all()
    assert np.array_equal(o0, o1) and np.array_equal(a0, a1)


@pytest.mark.parametrize("eq,shape", [(G.EQ_3D_RNGDEP, (7, 4, 61)), (G.EQ_GLOBAL_RNGDEP, (6, 9, 45))])
def test_fan_on_a_ragged_grid_is_variant_independent(eq, shape, monkeypatch):
    """a 96-ray fan over a ragged grid (rays cross cells of different sizes: the cell hint of grid_locate misses and rescans) gives the same
    records from every kernel variant - bit-identical where the corner sums associate the same way (eight / four lanes per ray with and without
    the record cache; one lane with per-lane gathers / cooperative LDS-DMA gather and forced sub-epochs), equal to rounding across those groups"""
    nx, ny, nz = shape
    spherical = eq == G.EQ_GLOBAL_RNGDEP
    x, y, z, T, u, v, rho = _synthetic_grid(nx, ny, nz, spherical, seed=11)
    if spherical:
        src = (0.0, float(np.degrees(x[nx // 2])), float(np.degrees(y[ny // 2])))
    else:
        src = (float(x[nx // 2]), float(y[ny // 2]), 0.0)
    th, ph = G.fan_enumerate(theta_min=2.0, theta_max=46.0, theta_step=4.0, phi_min=-180.0, phi_max=135.0, phi_step=45.0)

    def run(env):
        ctx = G.FanContext(eq, device=0, options=env)            # launch-plan options of this context (geoac_set_option; the GEOAC_ prefix of a key is optional)
        ctx.upload_atmo_3d(x, y, z, T, u, v, rho)
        ctx.set_params(bounces=1, calc_amp=1, mode=0, src=src)
        out = ctx.run(th, ph)
        ctx.close()
        return out
    ref, steps = run({})
    assert steps > 1000 and np.isfinite(ref).all()
    # the same association of the corner sums: the same bits
    for env in ({"GEOAC_OCT": "0"}, {"GEOAC_QUAD_CACHE": "0"}):
        rec, st = run(env)
        assert st == steps, env
        assert np.array_equal(rec, ref), env
    one, st1 = run({"GEOAC_GRID_LANES": "1", "GEOAC_SPREAD": "1", "GEOAC_GRID_COOP": "0"})
    sub, st2 = run({"GEOAC_GRID_LANES": "1", "GEOAC_SPREAD": "1", "GEOAC_SUB_MIN_WAVES": "0", "GEOAC_S_ROWS": "256"})
    assert st1 == st2 and np.array_equal(one, sub)
    # the hand-off's time-out path: with SUB_TEST_STALL the workgroups of sub-epoch 0 never publish their flag; the waiters give up, the library
    # repeats the fan without sub-epochs (a loop in geoac_fan_launch), returns the same records and reports the fallback through geoac_fan_status
    ctx = G.FanContext(eq, device=0, options={"GEOAC_GRID_LANES": "1", "GEOAC_SPREAD": "1", "GEOAC_SUB_MIN_WAVES": "0", "GEOAC_S_ROWS": "256", "SUB_TEST_STALL": "1"})
    ctx.upload_atmo_3d(x, y, z, T, u, v, rho)
    ctx.set_params(bounces=1, calc_amp=1, mode=0, src=src)
    stalled, st4 = ctx.run(th, ph)
    assert ctx.fan_status() & G.FAN_SUB_FALLBACK, "the stalled hand-off was not reported"
    assert st4 == st1 and np.array_equal(stalled, one)
    again, st5 = ctx.run(th, ph)                                   # the context keeps sub-epochs off and keeps saying so
    assert st5 == st1 and np.array_equal(again, one) and ctx.fan_status() & G.FAN_SUB_FALLBACK
    ctx.close()
    plain = G.FanContext(eq, device=0, options={"GEOAC_GRID_LANES": "1", "GEOAC_SPREAD": "1", "GEOAC_SUB_MIN_WAVES": "0", "GEOAC_S_ROWS": "256"})
    plain.upload_atmo_3d(x, y, z, T, u, v, rho); plain.set_params(bounces=1, calc_amp=1, mode=0, src=src); plain.run(th, ph)
    assert plain.fan_status() == 0
    plain.close()
    # one, two and four lanes per ray add the four corners in different orders: equal to rounding
    variants = [(one, st1)]
    if G.has_ab_kernels():                                       # the two-lane kernels are part of A/B builds only (make AB=1)
        variants.append(run({"GEOAC_GRID_LANES": "2"}))
    else:
        with pytest.raises(G.GeoAcError, match="A/B builds only"):
            run({"GEOAC_GRID_LANES": "2"})
    for rec, st in variants:
        assert st == steps
        assert np.array_equal(rec[..., 0:3], ref[..., 0:3])                      # VALID / STEPS / BROKE columns
        np.testing.assert_allclose(rec, ref, rtol=1e-7, atol=1e-9)
