"""CPU suite: the plain-C oracle against the golden vectors generated from the compiled reference
(tests/golden/make_golden.py).  Same compiler + libm + operand order => bit-for-bit."""
import numpy as np
import pytest

import harness as H

EQS = [H.EQ_GLOBAL, H.EQ_3D, H.EQ_2D]


@pytest.fixture(scope="module")
def oracles():
    return {eq: H.Oracle(eq) for eq in EQS}


@pytest.mark.parametrize("eq", EQS)
def test_tables_bitexact(eq, golden, oracles):
    g = golden(eq)
    t = oracles[eq].tables()
    for k in t:
        assert np.array_equal(t[k], g[f"tab_{k}"]), k


@pytest.mark.parametrize("eq", EQS)
def test_atmo_probe_bitexact(eq, golden, oracles):
    g = golden(eq)
    o9, rho = oracles[eq].atmo_probe(g["probe_x"])
    assert np.array_equal(o9, g["probe_out9"])
    assert np.array_equal(rho, g["probe_rho"])


@pytest.mark.parametrize("eq", EQS)
def test_absorption_probe_bitexact(eq, golden, oracles):
    g = golden(eq)
    a = oracles[eq].absorption_probe(g["abs_x"], g["abs_f"], 0.0, 0.3)
    assert np.array_equal(a, g["abs_alpha"])


@pytest.mark.parametrize("eq", EQS)
@pytest.mark.parametrize("amp", [1, 0])
@pytest.mark.parametrize("mode", [0, 1, 3])
def test_fan_records_bitexact(eq, amp, mode, golden, oracles):
    g = golden(eq)
    cfg = H.make_cfg(eq, calc_amp=bool(amp), mode=mode)
    want_smp = (amp == 1 and mode == 3)
    steps, rec, smp, nsmp = oracles[eq].fan(cfg, g["theta"], g["phi"], smp_cap=40000 if want_smp else 0)
    tag = f"amp{amp}_mode{mode}"
    assert steps == int(g[f"steps_{tag}"])
    assert np.array_equal(rec, g[f"rec_{tag}"])
    if want_smp:
        assert nsmp == int(g[f"nsmp_{tag}"])
        assert np.array_equal(smp[g[f"smp_idx_{tag}"]], g[f"smp_{tag}"])


@pytest.mark.parametrize("eq", EQS)
def test_fan_alt_config_bitexact(eq, golden, oracles):
    g = golden(eq)
    cfg = H.make_cfg(eq, bounces=int(g["altcfg_bounces"]), calc_amp=True, mode=0, src=tuple(g["altcfg_src"]),
                     z_grnd=float(g["altcfg_z_grnd"]), tweak_abs=float(g["altcfg_tweak_abs"]),
                     freq=float(g["altcfg_freq"]), range_limit=float(g["altcfg_range_limit"]))
    steps, rec, _, _ = oracles[eq].fan(cfg, g["theta"], g["phi"])
    assert steps == int(g["steps_alt"])
    assert np.array_equal(rec, g["rec_alt"])


@pytest.mark.parametrize("eq", EQS)
def test_stepper_rows_bitexact(eq, golden, oracles):
    g = golden(eq)
    cfg = H.make_cfg(eq, calc_amp=True)
    k, rows = oracles[eq].trace_leg0(cfg, 15.0, -90.0)
    assert k == int(g["trace_k"])
    assert np.array_equal(rows[g["trace_idx"]], g["trace_rows"])


def test_fan_enumeration_matches_reference_loop():
    # 0.1-degree steps give 449 rays, not 450, under repeated addition
    th, ph = H.fan_angles(theta_min=0.1, theta_max=45.0, theta_step=0.1)
    assert len(th) == 449
    th, ph = H.fan_angles(phi_min=-180.0, phi_max=179.0, phi_step=1.0)
    assert len(th) == 360 * 90
