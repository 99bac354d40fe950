"""CPU suite: the plain-C oracle of the range-dependent spherical set (GeoAcGlobal.RngDep: lat/lon grid of vertical splines,
quirks Q12 kept) against golden vectors generated from the compiled reference (make_golden.py globalrd)."""
import numpy as np
import pytest

import harness as H
import rngdep_data as RD

EQ = H.EQ_GLOBAL_RNGDEP


@pytest.fixture(scope="module")
def gold():
    return np.load(f"{H.GOLDEN_DIR}/globalrd_small.npz")


@pytest.fixture(scope="module")
def oracle(tmp_path_factory):
    d = tmp_path_factory.mktemp("gg")
    O = H.Oracle(EQ, met=None)
    O.load_grid(*RD.write_grid_global(str(d), short_paths=False))
    return O


def test_grid_interpolant_bitexact(gold, oracle):
    o30, a8 = oracle.grid_probe(gold["probe_r"], gold["probe_lat"], gold["probe_lon"])
    assert np.array_equal(o30, gold["probe_out30"])          # Eval_Spline_AllOrder2 of T, u, v
    assert np.array_equal(a8, gold["probe_api8"])            # c, rho, u, v, c_diff(r), u_diff(r), v_diff(r), c_diff(lat)


@pytest.mark.parametrize("amp,mode", [(1, 0), (0, 0), (1, 3)])
def test_fan_records_bitexact(gold, oracle, amp, mode):
    cfg = H.make_cfg(EQ, bounces=1, calc_amp=bool(amp), mode=mode, src=(0.0, 31.0, 0.0))
    want_smp = (amp == 1 and mode == 3)
    steps, rec, smp, nsmp = oracle.fan(cfg, gold["theta"], gold["phi"], smp_cap=40000 if want_smp else 0)
    tag = f"amp{amp}_mode{mode}"
    assert steps == int(gold[f"steps_{tag}"])
    assert np.array_equal(rec, gold[f"rec_{tag}"])
    if want_smp:
        assert nsmp == int(gold[f"nsmp_{tag}"])
        assert np.array_equal(smp[gold[f"smp_idx_{tag}"]], gold[f"smp_{tag}"])


def test_alt_config_bitexact(gold):
    """fresh context: the grid is loaded with the z_grnd the fan then uses (what -interactive / -eig_* do)"""
    import tempfile, os
    O = H.Oracle(EQ, met=None)
    O.load_grid(*RD.write_grid_global(os.path.join(tempfile.mkdtemp(), "a"), short_paths=False), z_grnd=0.0)
    cfg = H.make_cfg(EQ, bounces=2, calc_amp=True, mode=0, src=(1.5, 29.0, 1.0), z_grnd=0.3, freq=0.4, tweak_abs=0.6,
                     xy_limits=tuple(np.radians([26.0, 36.5, -7.0, 6.0])))
    steps, rec, _, _ = O.fan(cfg, gold["theta"], gold["phi"])
    assert steps == int(gold["steps_alt"])
    assert np.array_equal(rec, gold["rec_alt"])
