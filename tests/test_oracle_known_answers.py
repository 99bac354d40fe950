"""CPU suite: the analytic known-answer checks of tests/known_answers.py applied to the ORACLE (the plain-C restatement of the reference) -
straight rays in an isothermal windless medium for the three stratified sets, GeoAc3D = GeoAc2D without wind, and the reference's own
Hamiltonian residuals at arrivals.  The same closed forms judge the HIP path in tests/test_gpu_known_answers.py."""
import numpy as np
import pytest

import harness as H
import known_answers as K

STRAIGHT = [(H.EQ_3D, (3.0, -4.0, 20.0)), (H.EQ_2D, (20.0, 0.0, 0.0)), (H.EQ_GLOBAL, (20.0, 30.0, 10.0))]
TH = np.array([-40.0, -25.0, -10.0, -3.0, 5.0, 30.0])
PH = np.array([20.0, 160.0, -75.0, 110.0, 0.0, 45.0])


@pytest.mark.parametrize("eq,src", STRAIGHT)
def test_isothermal_windless_rays_are_straight(eq, src):
    z, T, u, v, rho = K.isothermal_profile()
    O = H.Oracle(eq, met=None)
    O.load_arrays(z, T, u, v, rho)                         # (heights: the oracle adds the Earth radius for the spherical set)
    steps, rec, _, _ = O.fan(H.make_cfg(eq, bounces=0, calc_amp=True, src=src), TH, PH)
    n, off, et, ea = K.check_straight_rays(eq, rec, TH, PH, src)
    print(H.EQ_NAMES[eq], f"{n} arrivals: off the launch line {off:.2e}, travel time {et:.2e}, amplitude vs spherical spreading {ea:.2e}")


def test_3d_equals_2d_without_wind():
    a = H.Oracle(H.EQ_3D).tables()
    z, T, rho = a["x"], a["T"], a["rho"]
    zero = np.zeros_like(z)
    th = np.array([2.0, 9.0, 17.0, 28.0, 41.0]); az = 37.0
    O2 = H.Oracle(H.EQ_2D, met=None); O2.load_arrays(z, T, zero, zero, rho)
    O3 = H.Oracle(H.EQ_3D, met=None); O3.load_arrays(z, T, zero, zero, rho)
    _, r2, _, _ = O2.fan(H.make_cfg(H.EQ_2D, bounces=1, calc_amp=False), th, np.full_like(th, az))
    _, r3, _, _ = O3.fan(H.make_cfg(H.EQ_3D, bounces=1, calc_amp=False), th, np.full_like(th, az))
    n, worst = K.check_2d_equals_3d_without_wind(r2, r3)
    print(f"{n} arrivals, GeoAc2D vs GeoAc3D without wind: worst relative difference {worst:.2e}")


def test_hamiltonian_residuals_at_arrivals_global():
    O = H.Oracle(H.EQ_GLOBAL)
    th, ph = H.fan_angles(theta_min=3.0, theta_max=43.0, theta_step=8.0, phi_min=-150.0, phi_max=150.0, phi_step=100.0)
    _, rec, _, _ = O.fan(H.make_cfg(H.EQ_GLOBAL, bounces=1, calc_amp=True), th, ph)
    c_src = O.atmo_probe(np.array([K.R_EARTH]))[0][0, 0]
    n, h, hd = K.hamiltonian_residuals(H.EQ_GLOBAL, rec, lambda x: O.atmo_probe(x)[0], c_src)
    n0, h0, hd0 = K.hamiltonian_residuals(H.EQ_GLOBAL, rec[:, :1], lambda x: O.atmo_probe(x)[0], c_src)
    print(f"{n} arrivals: |H| <= {h:.2e}; first legs: |H_deriv| / |mu| <= {hd0:.2e}, all legs {hd:.2e}")
    # (the derivative residual is 2e-3 on the first leg and grows with every reflection - the reference's reflection conditions for the auxiliary variables are approximate;
    #  a gross-error bound on the first leg, not an accuracy claim)
    assert n >= 10 and h < 1e-4 and hd0 < 2e-2
