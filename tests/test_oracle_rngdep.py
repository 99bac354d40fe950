"""CPU suite: the plain-C oracle of the range-dependent Cartesian set (bicubic-of-vertical-splines interpolant +
6/18-equation system) against golden vectors generated from the compiled reference (make_golden.py rngdep)."""
import numpy as np
import pytest

import harness as H
import rngdep_data as RD

EQ = H.EQ_3D_RNGDEP


@pytest.fixture(scope="module")
def gold():
    return np.load(f"{H.GOLDEN_DIR}/3drd_small.npz")


@pytest.fixture(scope="module")
def oracle(tmp_path_factory):
    d = tmp_path_factory.mktemp("gd")
    O = H.Oracle(EQ, met=None)
    O.load_grid(*RD.write_grid(str(d)))
    return O


def test_grid_interpolant_bitexact(gold, oracle):
    o30, a8 = oracle.grid_probe(gold["probe_x"], gold["probe_y"], gold["probe_z"])
    assert np.array_equal(o30, gold["probe_out30"])          # Eval_Spline_AllOrder2 of T, u, v
    assert np.array_equal(a8, gold["probe_api8"])            # c, rho, u, v, c_diff, u_diff, v_diff (Q11 forms)


@pytest.mark.parametrize("amp,mode", [(1, 0), (0, 0), (1, 3)])
def test_fan_records_bitexact(gold, oracle, amp, mode):
    cfg = H.make_cfg(EQ, bounces=1, calc_amp=bool(amp), mode=mode, src=(0.0, 0.0, 0.0))
    want_smp = (amp == 1 and mode == 3)
    steps, rec, smp, nsmp = oracle.fan(cfg, gold["theta"], gold["phi"], smp_cap=40000 if want_smp else 0)
    tag = f"amp{amp}_mode{mode}"
    assert steps == int(gold[f"steps_{tag}"])
    assert np.array_equal(rec, gold[f"rec_{tag}"])
    if want_smp:
        assert nsmp == int(gold[f"nsmp_{tag}"])
        assert np.array_equal(smp[gold[f"smp_idx_{tag}"]], gold[f"smp_{tag}"])
