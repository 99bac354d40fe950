"""Analytic known-answer checks (the reference has no tests; these are the ones it suggests): records of a fan in media whose rays are known
in closed form, and the reference's own unused self-checks (GeoAc_EvalHamiltonian / GeoAc_EvalHamiltonian_Deriv, EquationSets.Global.cpp:447-495,
3DStratified.cpp:315-321) evaluated at every arrival.  The checks take record tables - tests/test_gpu_known_answers.py feeds them the HIP path's,
tests/test_oracle_known_answers.py (CPU suite) the oracle's: the same closed forms judge both."""
import numpy as np

from harness import EQ_2D, EQ_3D, EQ_GLOBAL, REC

GAM_R = 0.00040187            # c = sqrt(gamR T)  (G2S_Spline1D.cpp:332)
R_EARTH = 6370.0


def isothermal_profile(T0=250.0, top=140.0, dz=0.1, scale_height=8.0, rho0=1.2e-3):
    """z, T, u, v, rho: constant temperature, no wind, exponential density - every ray is a straight line (a chord, in the spherical set)"""
    z = np.arange(0.0, top + 0.5 * dz, dz)
    return z, np.full_like(z, T0), np.zeros_like(z), np.zeros_like(z), rho0 * np.exp(-z / scale_height)


def check_straight_rays(eq, rec, theta, phi, src, T0=250.0, scale_height=8.0, rtol=1e-9, amp_rtol=1e-7):
    """isothermal windless medium, elevated source: rays launched downward arrive on the ground along their launch direction (Cartesian sets; the
    spherical set: in their great-circle plane, see below).
    Every arrival: the end point lies on the launch line (offset from it <= rtol x length), the slowness vector is the launch direction,
    travel time = length / c, and (Cartesian sets) the amplitude is spherical spreading 1 / (4 pi R) x sqrt(rho_arrival / rho_source)
    (2-D set: the reference takes rho at the ground for the source's: factor 1).  Returns (arrivals, worst offset, worst time error, worst amplitude error)."""
    c = np.sqrt(GAM_R * T0)
    th, az = np.radians(theta), np.radians(phi)
    valid = rec[:, 0, REC["VALID"]] > 0
    assert valid.sum() >= 3 and (rec[~valid, 0, REC["BROKE"]] > 0).all()
    st = rec[:, 0, REC["STATE"]:REC["STATE"] + 6]
    if eq == EQ_3D:
        d = np.stack([np.cos(th) * np.sin(az), np.cos(th) * np.cos(az), np.sin(th)], axis=1)
        p0 = np.array(src, dtype=float)
        p = st[:, 0:3]
        nu = np.stack([d[:, 0], d[:, 1], st[:, 3]], axis=1)          # nu_x, nu_y are constants of the ray: only nu_z is state
    elif eq == EQ_2D:
        d = np.stack([np.cos(th), np.sin(th)], axis=1)
        p0 = np.array([0.0, src[0]])
        p = st[:, 0:2]
        nu = np.stack([d[:, 0], st[:, 2]], axis=1)
    else:
        # Spherical set: the reference's slowness equations (Global.cpp:263-269, 385-390) are not those of straight lines in a homogeneous medium - its
        # rays bend towards the ground (d nu_r / ds = -(nu_t c_t + nu_p c_p) / (r |c_g|): the sign a rotating local frame asks for is +) - and
        # parity means following the reference.  What symmetry still dictates: the ray never leaves the plane through the Earth's centre that holds its
        # launch direction - its ground track is a GREAT CIRCLE - and the eikonal |nu| = c0 / c = 1 holds along it.
        z0, lat0, lon0 = src[0], np.radians(src[1]), np.radians(src[2])
        up = np.array([np.cos(lat0) * np.cos(lon0), np.cos(lat0) * np.sin(lon0), np.sin(lat0)])
        north = np.array([-np.sin(lat0) * np.cos(lon0), -np.sin(lat0) * np.sin(lon0), np.cos(lat0)])
        east = np.array([-np.sin(lon0), np.cos(lon0), 0.0])
        hdir = np.cos(az)[:, None] * north + np.sin(az)[:, None] * east
        normal = np.cross(np.broadcast_to(up, hdir.shape), hdir)        # of the great-circle plane
        r, la, lo = st[:, 0], st[:, 1], st[:, 2]
        p = np.stack([r * np.cos(la) * np.cos(lo), r * np.cos(la) * np.sin(lo), r * np.sin(la)], axis=1)
        u_ = np.stack([np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)], axis=1)
        n_ = np.stack([-np.sin(la) * np.cos(lo), -np.sin(la) * np.sin(lo), np.cos(la)], axis=1)
        e_ = np.stack([-np.sin(lo), np.cos(lo), np.zeros_like(lo)], axis=1)
        nu = st[:, 3:4] * u_ + st[:, 4:5] * n_ + st[:, 5:6] * e_      # slowness (nu_r, nu_lat, nu_lon) in the local frame of the END point -> ECEF
        L = np.linalg.norm(p - (R_EARTH + z0) * up, axis=1)[valid]
        off = np.abs((p * normal).sum(axis=1))[valid] / L              # distance of the end point from the plane, in path lengths
        assert off.max() <= rtol, f"end points off the great-circle plane by {off.max():.3e} of the path length"
        offn = np.abs((nu * normal).sum(axis=1))[valid]
        assert offn.max() <= rtol, f"slowness out of the great-circle plane by {offn.max():.3e}"
        eik = np.abs(np.linalg.norm(nu, axis=1)[valid] - 1.0)
        assert eik.max() <= 1e-7, f"|nu| differs from c0 / c = 1 by {eik.max():.3e}"
        return int(valid.sum()), float(off.max()), float(eik.max()), 0.0
    dp = (p - p0)[valid]
    L = np.linalg.norm(dp, axis=1)
    off = np.linalg.norm(dp - (dp * d[valid]).sum(axis=1, keepdims=True) * d[valid], axis=1) / L
    assert off.max() <= rtol, f"end points off the launch line by {off.max():.3e} of the path length"
    dnu = np.abs(nu[valid] - d[valid]).max()
    assert dnu <= rtol, f"slowness differs from the launch direction by {dnu:.3e}"
    tt = rec[:, 0, REC["TTIME"]][valid]
    if eq == EQ_2D:
        # GeoAc2D sums the segments 0 .. k-2 only (Q7: GeoAc2D_main.cpp:190-192): its travel time stops one step - one to three metres at the ground - short
        short = L - tt * c
        assert (short > 0.0).all() and (short <= 0.003).all(), f"2-D travel time is not the path less its last step: {short}"
        et = np.zeros(1)
    else:
        et = np.abs(tt * c / L - 1.0)
        assert et.max() <= rtol, f"travel time x c differs from the path length by {et.max():.3e}"
    ea = np.zeros(1)
    if eq in (EQ_3D, EQ_2D):
        z_src = src[2] if eq == EQ_3D else src[0]
        dens = np.exp(0.5 * z_src / scale_height) if eq == EQ_3D else 1.0   # sqrt(rho(ground) / rho(source)); below-ground samples are clamped to rho(0)
        want = dens / (4.0 * np.pi * L)
        ea = np.abs(rec[:, 0, REC["AMP"]][valid] / want - 1.0)
        assert ea.max() <= amp_rtol, f"amplitude differs from spherical spreading by {ea.max():.3e}"
    return int(valid.sum()), float(off.max()), float(et.max()), float(ea.max())


def check_2d_equals_3d_without_wind(rec2, rec3, rtol=2e-6):
    """windless stratified medium: GeoAc3D along any azimuth traces the rays of GeoAc2D.  The two sets step along different parameters (2-D: dr/ds = c / c0 cos(theta),
    not arc length; 2DStratified.cpp:152-181 against 3DStratified.cpp:251-310), so they agree to the accuracy of the RK4 scheme, not to rounding: the same legs
    arrive, range / vertical slowness / travel time / attenuation / turning height within rtol (measured 1e-7)"""
    assert np.array_equal(rec2[..., REC["VALID"]], rec3[..., REC["VALID"]]) and np.array_equal(rec2[..., REC["BROKE"]], rec3[..., REC["BROKE"]])
    valid = rec2[..., REC["VALID"]] > 0
    assert valid.sum() > 0
    s2, s3 = rec2[..., REC["STATE"]:REC["STATE"] + 3][valid], rec3[..., REC["STATE"]:REC["STATE"] + 4][valid]
    worst = 0.0
    for a, b, what in ((s2[:, 0], np.hypot(s3[:, 0], s3[:, 1]), "range"), (s2[:, 2], s3[:, 3], "nu_z"),
                       (rec2[..., REC["TTIME"]][valid], rec3[..., REC["TTIME"]][valid], "travel time"),
                       (rec2[..., REC["ATTEN"]][valid], rec3[..., REC["ATTEN"]][valid], "attenuation"),
                       (rec2[..., REC["TURN"]][valid], rec3[..., REC["TURN"]][valid], "turning height")):
        # (GeoAc2D's sums stop one step short of the arrival, Q7: one step of ~1e-3 km in ~1e2 .. 1e3 km of path - inside rtol x 10 for the two integrals)
        tol = rtol * (10.0 if what in ("travel time", "attenuation") else 1.0)
        e = np.abs(a - b) / np.maximum(np.abs(b), 1e-12)
        assert e.max() <= tol, f"{what}: 2-D and 3-D differ by {e.max():.3e}"
        worst = max(worst, float(e.max()))
    return int(valid.sum()), worst


def hamiltonian_residuals(eq, rec, atmo9_at, c_src):
    """the reference's self-checks at every arrival of a fan with amplitudes, stratified Global set.
    GeoAc_EvalHamiltonian: |nu| - c0 / c + nu . wind / c; GeoAc_EvalHamiltonian_Deriv: for each launch-angle system (R, mu)
    nu . mu / |nu| + |nu| / c c' R_z + (mu . wind + nu . wind' R_z) / c (medium depends on height only).  atmo9_at(x) -> rows of c, c', c'', u, u', u'', v, v', v''
    at the height coordinate x.  Returns (arrivals, max |H|, max |H_deriv| relative to |mu|)."""
    valid = rec[..., REC["VALID"]] > 0
    st = rec[..., REC["STATE"]:REC["STATE"] + 18][valid]
    if eq != EQ_GLOBAL:
        raise ValueError("hamiltonian_residuals: stratified Global set only (the 3-D set keeps nu_x, nu_y outside its state)")
    a = atmo9_at(st[:, 0])
    c, dc, u, du, v, dv = a[:, 0], a[:, 1], a[:, 3], a[:, 4], a[:, 6], a[:, 7]
    nu = st[:, 3:6]                                                   # nu_r, nu_lat, nu_lon; winds (w, v, u) -> (r, lat, lon)
    wind, dwind = np.stack([0 * u, v, u], axis=1), np.stack([0 * u, dv, du], axis=1)
    systems = [(st[:, 6], st[:, 9:12]), (st[:, 12], st[:, 15:18])]    # (R_r, mu)
    mag = np.linalg.norm(nu, axis=1)
    H = mag - c_src / c + (nu * wind).sum(axis=1) / c
    worst_d = 0.0
    for Rz, mu in systems:
        res = (nu * mu).sum(axis=1) / mag + mag / c * dc * Rz + ((mu * wind).sum(axis=1) + (nu * dwind).sum(axis=1) * Rz) / c
        worst_d = max(worst_d, float((np.abs(res) / np.maximum(np.linalg.norm(mu, axis=1), 1e-30)).max()))
    return int(valid.sum()), float(np.abs(H).max()), worst_d
