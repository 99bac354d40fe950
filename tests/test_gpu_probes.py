"""Device-function probes (include/geoac_probe.h) against the reference's own function values (golden vectors made by the compiled
reference, tests/golden/make_golden.py): the 1-D spline accessors, SuthBass_Alpha over 0.01 - 10 Hz (a13) and the
grid interpolants of the two range-dependent sets (a7, a8; per-lane gathers and the wave-cooperative gather) are checked point by point,
through the very device functions the RK4 and post-pass kernels call - not only through whole-fan integrals."""
import numpy as np
import pytest

import harness as H
from geoac_amd.api import DEFAULT_OPTIONS as OPT      # launch-plan options of the contexts the tests create (geoac_set_option)

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def _colwise(got, want, floor=1e-6, field_scale=None):
    """max over points of |got - want| / max(|want|, floor x column scale): 1e-6 relative, a value that passes through zero judged
    against 1e-6 of its column's scale.  field_scale (per column: the magnitude of the FIELD the column is a derivative of): a
    derivative that vanishes identically for the synthetic atmosphere (u_xx of a wind that is linear in x: 1e-20 of rounding noise in
    the reference and here) is judged against 1e-7 of its field (absolute error 1e-13 x the field) instead of against its own noise."""
    scale = np.maximum(np.abs(want).max(axis=0, keepdims=True), 1e-300)
    den = np.maximum(np.abs(want), floor * scale)
    if field_scale is not None:
        den = np.maximum(den, 1e-7 * np.asarray(field_scale)[None, :])
    return (np.abs(got - want) / den).max(axis=0)


def _field_scales(want30):
    return np.repeat(np.abs(want30[:, [0, 10, 20]]).max(axis=0), 10)


def _ctx_1d(eq):
    import geoac_amd as G
    ctx = G.FanContext(eq, device=0)
    ctx.load_met(H.TOYATMO)
    ctx.set_params(bounces=0, calc_amp=1, mode=0)
    ctx.run(np.array([20.0]), np.array([-90.0]))                 # the probes use the parameter block of a completed launch
    return ctx


@pytest.mark.parametrize("eq", [H.EQ_GLOBAL, H.EQ_3D, H.EQ_2D])
def test_spline_accessors_vs_reference(eq):
    """c, c', c'', u, u', u'', v, v', v'', rho at 1000 abscissae (nodes, both ends, beyond both ends): G2S_GlobalSpline1D.cpp:216-295,
    332-428 / G2S_Spline1D.cpp:245-281, 321-416"""
    g = np.load(f"{H.GOLDEN_DIR}/{H.EQ_NAMES[eq]}_small.npz")
    ctx = _ctx_1d(eq)
    o9, rho = ctx.probe_atmo_1d(g["probe_x"])
    e = _colwise(o9, g["probe_out9"])
    er = _colwise(rho[:, None], g["probe_rho"][:, None])
    print(H.EQ_NAMES[eq], "out9 max rel err per column", [f"{v:.1e}" for v in e], "rho", f"{er[0]:.1e}")
    assert e.max() <= RTOL and er.max() <= RTOL


@pytest.mark.parametrize("eq", [H.EQ_GLOBAL, H.EQ_3D, H.EQ_2D])
def test_absorption_vs_reference(eq):
    """SuthBass_Alpha at 200 (altitude, frequency) pairs, 0.01 - 10 Hz (Atmo_State.Absorption{,.Global}.cpp:12-141): the most heavily
    rewritten function of the device code (reciprocal / rsq forms, own exp / exp10), on its own"""
    g = np.load(f"{H.GOLDEN_DIR}/{H.EQ_NAMES[eq]}_small.npz")
    ctx = _ctx_1d(eq)
    a = ctx.probe_absorption(g["abs_x"], g["abs_f"])
    want = g["abs_alpha"]
    rel = np.abs(a - want) / np.abs(want)
    print(H.EQ_NAMES[eq], "alpha: f range", g["abs_f"].min(), g["abs_f"].max(), "max rel err", rel.max())
    assert (want > 0).all() and rel.max() <= RTOL


@pytest.mark.parametrize("coop", [False, True])
def test_cartesian_grid_interpolant_vs_reference(coop, tmp_path):
    """Eval_Spline_AllOrder2 of T, u, v (30 values) and the scalar API at 400 points of the 5x5 grid (Q11 included):
    G2S_MultiDimSpline3D.cpp:1341-1593, 1633-1743"""
    import geoac_amd as G
    import rngdep_data as RD
    g = np.load(f"{H.GOLDEN_DIR}/3drd_small.npz")
    ctx = G.FanContext(G.EQ_3D_RNGDEP, device=0)
    ctx.load_grid(*RD.write_grid(str(tmp_path), short_paths=False))
    ctx.set_params(bounces=0, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
    ctx.run(np.array([20.0]), np.array([-90.0]))
    o30, a7 = ctx.probe_grid(g["probe_x"], g["probe_y"], g["probe_z"], coop=coop)
    e = _colwise(o30, g["probe_out30"], field_scale=_field_scales(g["probe_out30"]))
    ea = _colwise(a7, g["probe_api8"][:, :7])
    print("3drd coop" if coop else "3drd", "AllOrder2 max rel err", f"{e.max():.1e}", "scalar API", [f"{v:.1e}" for v in ea])
    assert e.max() <= RTOL and ea.max() <= RTOL


@pytest.mark.parametrize("coop", [False, True])
def test_spherical_grid_interpolant_vs_reference(coop, tmp_path):
    """the spherical twin with its quirks Q12: G2S_GlobalMultiDimSpline3D.cpp:1224-1461, 1502-1611.  The reference orders its outputs
    f, r, t, p, rr, tt, pp, rt, rp, tp; the device table order is (lat, lon, r)"""
    import geoac_amd as G
    import rngdep_data as RD
    g = np.load(f"{H.GOLDEN_DIR}/globalrd_small.npz")
    ctx = G.FanContext(G.EQ_GLOBAL_RNGDEP, device=0)
    ctx.load_grid(*RD.write_grid_global(str(tmp_path), short_paths=False))
    ctx.set_params(bounces=0, calc_amp=1, mode=0, src=(0.0, 31.0, 0.0))
    ctx.run(np.array([20.0]), np.array([-90.0]))
    o30, a7 = ctx.probe_grid(g["probe_lat"], g["probe_lon"], g["probe_r"], coop=coop)
    dev_of_ref = [0, 3, 1, 2, 6, 4, 5, 8, 9, 7]                  # reference slot q <- device slot
    got = np.concatenate([o30[:, 10 * f + np.array(dev_of_ref)] for f in range(3)], axis=1)
    # the first six probe points sit exactly on grid nodes in all three coordinates: the patch is C1 there, but the reference's mixed
    # second derivatives (left in scaled cell coordinates, Q12) jump from cell to cell, and which cell the reference takes at an exact
    # node depends on the history of its search cursor (Q13) - those six points are compared on value and gradient only
    want = g["probe_out30"]
    second = np.array([q >= 4 for q in range(10)] * 3)
    e = _colwise(got[6:], want[6:], field_scale=_field_scales(want))
    e_nodes = _colwise(got[:6][:, ~second], want[:6][:, ~second], field_scale=_field_scales(want)[~second])
    e = np.concatenate([e, e_nodes])
    ea = _colwise(a7, g["probe_api8"][:, :7])
    print("globalrd coop" if coop else "globalrd", "AllOrder2 max rel err", f"{e.max():.1e}", "scalar API", [f"{v:.1e}" for v in ea])
    print("   per column", [f"{v:.0e}" for v in e])
    assert e.max() <= RTOL and ea.max() <= RTOL


@pytest.mark.parametrize("eqname", ["EQ_3D_RNGDEP", "EQ_GLOBAL_RNGDEP"])
def test_cooperative_and_per_lane_gathers_give_the_same_bits(eqname, tmp_path):
    """4096 random points of the grid through the per-lane evaluator (grid_eval_all) and through the wave-cooperative one (LDS-DMA ring for the
    Cartesian set, register-staged exchange for the spherical one): bit-identical - a ray's numbers must not depend on which gather served it"""
    import geoac_amd as G
    import rngdep_data as RD
    rng = np.random.default_rng(5)
    n = 4096
    ctx = G.FanContext(getattr(G, eqname), device=0)
    if eqname == "EQ_3D_RNGDEP":
        ctx.load_grid(*RD.write_grid(str(tmp_path), short_paths=False))
        ctx.set_params(bounces=0, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
        ctx.run(np.array([20.0]), np.array([-90.0]))
        a = (rng.uniform(-400, 400, n), rng.uniform(-400, 400, n), rng.uniform(0.0, 130.0, n))
    else:
        ctx.load_grid(*RD.write_grid_global(str(tmp_path), short_paths=False))
        ctx.set_params(bounces=0, calc_amp=1, mode=0, src=(0.0, 31.0, 0.0))
        ctx.run(np.array([20.0]), np.array([-90.0]))
        a = (np.radians(rng.uniform(27.0, 35.0, n)), np.radians(rng.uniform(-4.0, 4.0, n)), 6370.0 + rng.uniform(0.0, 130.0, n))
    o0, a0 = ctx.probe_grid(*a, coop=False)
    o1, a1 = ctx.probe_grid(*a, coop=True)
    assert np.isfinite(o0).all()
    assert np.array_equal(o0, o1) and np.array_equal(a0, a1)


@pytest.mark.parametrize("eq", [H.EQ_GLOBAL, H.EQ_3D, H.EQ_2D])
def test_absorption_table_vs_exact_routine_and_reference(eq):
    """The table the post-pass of the stratified sets reads (k_atab_build: degree-7 interpolant of SuthBass_Alpha per spline segment,
    Atmo_State.Absorption{,.Global}.cpp:12-141) against (a) the exact device routine at 40 000 abscissae - nodes, points next to nodes and
    the strips beyond both ends of the profile included - to 1e-9 relative, and (b) the reference's own values at the golden (altitude,
    frequency) pairs to 1e-6 (a table per frequency).  Points the table does not serve come back as -1: none inside ToyAtmo."""
    g = np.load(f"{H.GOLDEN_DIR}/{H.EQ_NAMES[eq]}_small.npz")
    ctx = _ctx_1d(eq)
    info = ctx.abs_table_info()
    import geoac_amd as G
    a = G.met_load(H.TOYATMO, eq)
    x0, x1, nodes = a["x"][0], a["x"][-1], a["x"]
    assert info["entries"] == len(nodes) + 1, info                   # nseg + 2
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.uniform(x0 - 0.04, x1 + 0.04, 36000), nodes, nodes[1:] - 1e-9, nodes[:-1] + 1e-9,
                         np.array([x0 - 0.049, x0 - 1e-12, x1 + 1e-12, x1 + 0.049, x0 - 0.2, x1 + 0.2])])
    tab = ctx.probe_absorption_table(xs)
    ex = ctx.probe_absorption(xs, np.full(len(xs), 0.1))
    served = tab >= 0.0
    beyond = (xs < x0 - 0.05) | (xs > x1 + 0.05)
    assert not served[beyond].any()                                   # beyond the strips: left to the exact pass
    inside = ~beyond
    print(H.EQ_NAMES[eq], "table entries", info["entries"], "flagged", info["flagged"], "served", served[inside].mean())
    assert info["flagged"] == 0 and served[inside].all()
    # An abscissa that IS a node belongs to two segments; where the node is also a branch point of the reference's piecewise fits
    # (30, 76, 80, 90, 95 km) the segment above it holds the upper fit and the exact routine takes the lower one AT the point (zr > 30. is
    # false there): the two values differ by the jump of the fit itself (0.5 % at 30 km).  A path-segment midpoint hits a node with
    # probability zero; the points a nanometre to either side must agree.
    at_node = np.isin(xs, nodes)
    rel = np.abs(tab - ex) / ex
    print(H.EQ_NAMES[eq], "table vs exact routine: max rel err off the nodes", rel[served & ~at_node].max(), "at the nodes", rel[served & at_node].max())
    assert rel[served & ~at_node].max() <= 1e-9                      # (the exact routine itself carries ~2e-11 of rounding noise from sqrt(1 + nu^2) - 1)
    assert rel[served & at_node].max() <= 1e-2 and np.median(rel[served & at_node]) <= 1e-12
    # the reference's values: one table per frequency
    idx = np.argsort(g["abs_f"])[np.linspace(0, len(g["abs_f"]) - 1, 12).astype(int)]
    worst = 0.0
    for i in idx:
        f = float(g["abs_f"][i])
        ctx.set_params(freq=f)
        ctx.run(np.array([20.0]), np.array([-90.0]))
        same_f = np.isclose(g["abs_f"], f, rtol=0, atol=0)
        t = ctx.probe_absorption_table(g["abs_x"][same_f])
        want = g["abs_alpha"][same_f]
        assert (t >= 0).all()
        worst = max(worst, float((np.abs(t - want) / want).max()))
    print(H.EQ_NAMES[eq], "table vs reference SuthBass_Alpha at", len(idx), "frequencies: max rel err", worst)
    assert worst <= RTOL


@pytest.mark.parametrize("eq", [H.EQ_GLOBAL, H.EQ_3D, H.EQ_2D])
def test_table_and_exact_post_pass_agree(eq, monkeypatch):
    """A fan through the table post-pass (k_postpass_tab) and through the exact one (GEOAC_ABS_TABLE=0): travel times identical,
    attenuations to 1e-10 relative; with a raised ground and a lowered ceiling some midpoints fall below the first / above the last node
    (strips) and rays leave through the top."""
    import geoac_amd as G
    th, ph = G.fan_enumerate(theta_min=1.0, theta_max=45.0, theta_step=1.0, phi_min=-90.0, phi_max=0.0, phi_step=30.0)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setitem(OPT, "ABS_TABLE", mode)
        ctx = G.FanContext(eq, device=0)
        ctx.load_met(H.TOYATMO)
        ctx.set_params(bounces=2, calc_amp=1, mode=0)
        rec, steps = ctx.run(th, ph)
        out[mode] = (rec.copy(), steps, ctx.abs_table_info())
        ctx.close()
    (r1, s1, i1), (r0, s0, i0) = out["1"], out["0"]
    assert s1 == s0 and i1["entries"] > 0 and i0["entries"] == 0
    valid = r0[:, :, G.REC["VALID"]] == 1.0
    tt1, tt0 = r1[:, :, G.REC["TTIME"]][valid], r0[:, :, G.REC["TTIME"]][valid]
    at1, at0 = r1[:, :, G.REC["ATTEN"]][valid], r0[:, :, G.REC["ATTEN"]][valid]
    print(H.EQ_NAMES[eq], "arrivals", valid.sum(), "fix-up segments", i1["fixup_segments"], "TTIME max rel", np.abs(tt1 / tt0 - 1).max(), "ATTEN max rel", np.abs(at1 / at0 - 1).max())
    assert np.abs(tt1 / tt0 - 1).max() <= 1e-14
    assert np.abs(at1 / at0 - 1).max() <= 1e-10
    other = [k for k in G.REC if k not in ("TTIME", "ATTEN")]
    for k in other:
        np.testing.assert_array_equal(r1[:, :, G.REC[k]], r0[:, :, G.REC[k]])


@pytest.mark.parametrize("cap", [1 << 20, 64])
def test_segments_the_table_does_not_serve_go_through_the_fix_up_pass(cap, monkeypatch, capfd):
    """A table built with a tolerance tighter than the interpolants reach (ABS_TABLE_TOL) flags part of its entries; the path segments that fall into
    them are listed by k_postpass_tab and evaluated exactly by k_ppfix - the fan still agrees with the exact post-pass to 1e-10.  With a list of 64
    entries (PPFIX_CAP) the list overflows: the library repeats the fan with the exact post-pass and says so (geoac_fan_status: GEOAC_FAN_ABS_FALLBACK)."""
    import geoac_amd as G
    th, ph = G.fan_enumerate(theta_min=1.0, theta_max=45.0, theta_step=1.0, phi_min=-90.0, phi_max=0.0, phi_step=30.0)
    monkeypatch.setitem(OPT, "ABS_TABLE", "0")
    ctx = G.FanContext(H.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO); ctx.set_params(bounces=2, calc_amp=1, mode=0)
    r0, s0 = ctx.run(th, ph); r0 = r0.copy(); ctx.close()
    monkeypatch.setitem(OPT, "ABS_TABLE", "1")
    got = None
    for tol in ("2e-11", "1e-11", "5e-12", "2e-12"):                 # (the spherical set's worst check-point error is ~3e-11: some tolerance below it flags a part, not all)
        monkeypatch.setitem(OPT, "ABS_TABLE_TOL", tol); monkeypatch.setitem(OPT, "PPFIX_CAP", str(cap))
        ctx = G.FanContext(H.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO); ctx.set_params(bounces=2, calc_amp=1, mode=0)
        r1, s1 = ctx.run(th, ph); info = ctx.abs_table_info(); status = ctx.fan_status(); r1 = r1.copy(); ctx.close()
        if cap == 64 and info["entries"] == 0 and (status & G.FAN_ABS_FALLBACK):
            got = (tol, r1, s1, info); break                         # (the overflow was reported and the fan repeated without the table)
        if info["entries"] and 0 < info["flagged"] and info["fixup_segments"] > 0:
            got = (tol, r1, s1, info); break
    assert got is not None, "no tolerance flagged a part of the table"
    tol, r1, s1, info = got
    print("tolerance", tol, info)
    assert s1 == s0
    valid = r0[:, :, G.REC["VALID"]] == 1.0
    at1, at0 = r1[:, :, G.REC["ATTEN"]][valid], r0[:, :, G.REC["ATTEN"]][valid]
    assert np.abs(at1 / at0 - 1).max() <= 1e-10
    np.testing.assert_array_equal(r1[:, :, G.REC["TTIME"]], r0[:, :, G.REC["TTIME"]]) if info["entries"] == 0 else None



@pytest.mark.parametrize("eq", ["global", "3d"])
def test_one_trip_and_walking_forms_of_the_table_post_pass_give_the_same_bits(eq, monkeypatch):
    """k_postpass_tab finds the spline segment of a path segment's midpoint either by the hinted walk (hybrid fans) or by fetching the neighbouring
    record and table entry in one trip and walking only when that record does not hold the midpoint (PP_ONETRIP; default on the fans that fill
    the chip): the same segment, the same records bit for bit - steep rays included (up to 85 deg: a node every two or three steps)"""
    import geoac_amd as G
    th, ph = G.fan_enumerate(theta_min=1.0, theta_max=85.0, theta_step=2.0, phi_min=-90.0, phi_max=90.0, phi_step=45.0)
    EQ = H.EQ_GLOBAL if eq == "global" else H.EQ_3D
    out = {}
    for v in ("0", "1"):
        monkeypatch.setitem(OPT, "PP_ONETRIP", v)
        ctx = G.FanContext(EQ, device=0); ctx.load_met(H.TOYATMO); ctx.set_params(bounces=2, calc_amp=1, mode=0)
        r, s = ctx.run(th, ph); out[v] = (r.copy(), s, ctx.abs_table_info()); ctx.close()
    assert out["0"][2]["entries"] > 0 and out["1"][2]["entries"] > 0
    assert out["0"][1] == out["1"][1]
    assert np.array_equal(out["0"][0].view(np.uint64), out["1"][0].view(np.uint64))


@pytest.mark.parametrize("amp", [1, 0])
def test_table_entry_in_lds_gives_the_register_forms_bits(amp, monkeypatch):
    """k_postpass_tab<EqGlobal, one trip, TBL>: the table entry in hand lives in LDS (19 x 256 doubles per workgroup) instead of 38 registers and no row is
    prefetched - 127 registers, four waves per SIMD, the default of the spherical set's fans that fill the chip (PP_LDS_TABLE).  Same operations
    on the same operands as the register form: the same records bit for bit, steep rays (a node every two or three steps) and several legs included,
    also with the short chunks that cut a thread's sixteen-segment walk (S_ROWS)."""
    import geoac_amd as G
    th, ph = G.fan_enumerate(theta_min=1.0, theta_max=85.0, theta_step=2.0, phi_min=-90.0, phi_max=90.0, phi_step=45.0)
    out = {}
    for name, opts in (("reg", {"PP_ONETRIP": "1", "PP_LDS_TABLE": "0"}), ("lds", {"PP_ONETRIP": "1", "PP_LDS_TABLE": "1"}),
                       ("lds_short", {"PP_ONETRIP": "1", "PP_LDS_TABLE": "1", "S_ROWS": "1000"}), ("walk", {"PP_ONETRIP": "0"})):
        for k, v in opts.items():
            monkeypatch.setitem(OPT, k, v)
        ctx = G.FanContext(H.EQ_GLOBAL, device=0); ctx.load_met(H.TOYATMO); ctx.set_params(bounces=2, calc_amp=amp, mode=0)
        r, s = ctx.run(th, ph); out[name] = (r.copy(), s, ctx.abs_table_info()); ctx.close()
        for k in opts:
            monkeypatch.delitem(OPT, k)
    assert out["lds"][2]["entries"] > 0
    for name in ("lds", "lds_short", "walk"):
        assert out[name][1] == out["reg"][1]
        assert np.array_equal(out[name][0].view(np.uint64), out["reg"][0].view(np.uint64)), name
