"""GPU tests at BASELINE.json's full sizes.  The oracle cannot integrate these fans in seconds, so they are checked through
size-independent properties plus a random sample of rays re-integrated by the oracle:
  * ray independence: the records of a sampled ray do not depend on the fan it was part of (bitwise),
  * eikonal (Hamiltonian) residual at every arrival: |nu| = c0 / c(ground) (winds are tapered to 0 at the ground),
  * count bookkeeping: steps of all legs add up to the device counter; a leg that broke has no later legs,
  * cumulative travel time / attenuation are non-decreasing over the legs of a ray."""
import os

import numpy as np
import pytest

import harness as H
from parity import compare_records

pytestmark = pytest.mark.gpu

REC = H.REC


@pytest.fixture(scope="module")
def G():
    import geoac_amd
    geoac_amd.load_library()
    return geoac_amd


from parity import fan_properties as _properties  # noqa: E402


def _sample_vs_oracle(G, eq, ctx_params, theta, phi, rec, E, n=24, seed=3):
    rng = np.random.default_rng(seed)
    idx = np.sort(rng.choice(len(theta), n, replace=False))
    O = H.Oracle(eq)
    cfg = H.make_cfg(eq, bounces=ctx_params["bounces"], calc_amp=bool(ctx_params["calc_amp"]), mode=0)
    so, ro, _, _ = O.fan(cfg, theta[idx], phi[idx])
    compare_records(rec[idx], ro, E=E)
    # ray independence: the same rays as their own small fan, bitwise
    ctx = G.FanContext(eq, device=0)
    ctx.load_met(H.TOYATMO)
    ctx.set_params(**ctx_params)
    r2, s2 = ctx.run(theta[idx], phi[idx])
    assert s2 == so
    assert np.array_equal(r2, rec[idx])


def test_metric_fan_global_360x90(G):
    """BASELINE metric / config: GeoAcGlobal 360 az x 90 incl, bounces=2, CalcAmp=True (874 273 730 ray-steps)"""
    th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0)
    params = dict(bounces=2, calc_amp=1, mode=0)
    ctx = G.FanContext(G.EQ_GLOBAL, device=0)
    ctx.load_met(H.TOYATMO)
    ctx.set_params(**params)
    rec, steps = ctx.run(th, ph)
    assert len(th) == 32400
    c_g = H.Oracle(H.EQ_GLOBAL).atmo_probe(np.array([6370.0]))[0][0, 0]
    narr = _properties(rec, steps, 18, slice(3, 6), c_ratio=1.0)      # source on the ground: c0 = c(ground)
    assert steps == 874273730                                         # (the reference's own total: tests/golden/full_metric.npz, test_gpu_fullfan.py)
    assert narr > 90000
    _sample_vs_oracle(G, H.EQ_GLOBAL, params, th, ph, rec, E=18)


def test_config2_3d_360x90(G):
    """configs[1]: GeoAc3D stratified ToyAtmo, 360 az x 90 incl, CalcAmp=True"""
    th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0)
    params = dict(bounces=2, calc_amp=1, mode=0)
    ctx = G.FanContext(G.EQ_3D, device=0)
    ctx.load_met(H.TOYATMO)
    ctx.set_params(**params)
    rec, steps = ctx.run(th, ph)
    # 3-D set: nu_x, nu_y are ray constants, not in the state; check the vertical component bound only
    valid = rec[..., REC["VALID"]] > 0
    assert int(rec[..., REC["STEPS"]].sum()) == steps
    assert np.abs(rec[..., REC["STATE"] + 3][valid]).max() <= 1.0 + 1e-9
    _sample_vs_oracle(G, H.EQ_3D, params, th, ph, rec, E=12)


def test_config3_global_720x180_bounces3(G):
    """configs[2]: GeoAcGlobal 720 az x 180 incl, bounces=3 (129 600 rays x 4 legs)"""
    th, ph = G.fan_enumerate(theta_min=0.25, theta_max=45.0, theta_step=0.25, phi_min=-180.0, phi_max=179.5, phi_step=0.5)
    assert len(th) == 129600
    params = dict(bounces=3, calc_amp=1, mode=0)
    ctx = G.FanContext(G.EQ_GLOBAL, device=0)
    ctx.load_met(H.TOYATMO)
    ctx.set_params(**params)
    rec, steps = ctx.run(th, ph)
    _properties(rec, steps, 18, slice(3, 6), c_ratio=1.0)
    _sample_vs_oracle(G, H.EQ_GLOBAL, params, th, ph, rec, E=18, n=16)
    # staggered epochs (fans with more waves than the chip has wave slots: the shallow share of the compacted list gets a whole epoch's rows per launch, the rest fewer
    # on a second stream - the default plan here): off, and with another share and ratio - which columns got how many rows changes nothing in the records
    ref = rec.copy(); ctx.close()
    # (... and NO_PAIR: the last epochs stay on the one-lane kernel instead of changing to the two-lane one; ACCUM_BATCH=0: the sums row by row)
    for env in ({"GEOAC_STAGGER_FRAC": "0"}, {"GEOAC_STAGGER_FRAC": "0.3", "GEOAC_STAGGER_ROWS": "0.45"}, {"GEOAC_NO_PAIR": "1", "GEOAC_ACCUM_BATCH": "0"}):
        with G.options(**env):
            c2 = G.FanContext(G.EQ_GLOBAL, device=0)
            c2.load_met(H.TOYATMO); c2.set_params(**params)
            r2, s2 = c2.run(th, ph); c2.close()
        assert s2 == steps, env
        assert np.array_equal(r2.view(np.uint64), ref.view(np.uint64)), env


@pytest.mark.parametrize("eqname,total", [("EQ_GLOBAL", 874273730), ("EQ_3D", 871080426)])
def test_full_fan_is_schedule_independent(G, eqname, total):
    """the 360 x 90 fan of the spherical and of the 3-D stratified set under different launch plans - default hybrid split, a deliberately bad split (only 3 % of the rays on the
    two-lane kernel: the merge-back rule takes over), one lane for every ray with and without live-ray compaction between epochs, two chunks, 4096-row epochs - gives bit-identical
    records: which kernel variant integrates a ray, and in which epoch pattern, must not matter"""
    th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0)
    params = dict(bounces=2, calc_amp=1, mode=0)

    def run(env):
        with G.options(**env):                                # launch-plan options of the contexts created inside (geoac_set_option)
            ctx = G.FanContext(getattr(G, eqname), device=0)   # the knobs are read when the context is created
            ctx.load_met(H.TOYATMO)
            ctx.set_params(**params)
            rec, steps = ctx.run(th, ph)
            ctx.close()
        return rec, steps
    ref, steps = run({})
    assert steps == total
    # GEOAC_NO_PAIR: one launch per epoch, which runs over the compacted list of live rays (k_compact); with GEOAC_COMPACT=0 over all slots.
    # GEOAC_DUO=1 (spherical set): the wave-specialised kernel (k_rk4_duo: the ray on one wave, its derivative systems on another, the
    # stage values handed over through LDS) with and without compaction and with short epochs - same bits as the one-wave kernels.
    plans = [{"GEOAC_PAIR_FRAC": "0.03"}, {"GEOAC_NO_PAIR": "1"}, {"GEOAC_NO_PAIR": "1", "GEOAC_COMPACT": "0"},
             {"GEOAC_TWO_CHUNKS": "1", "GEOAC_S_ROWS": "4096"}, {"GEOAC_PAIR_FRAC": "1.0"}, {"GEOAC_PAIR_FRAC": "0"}]
    # k_accum with the contributions of eight rows fetched together in EVERY epoch / in none (default: the late epochs only), and small path chunks (CHUNK_GIB: a shared device)
    plans += [{"GEOAC_ACCUM_BATCH": "1"}, {"GEOAC_ACCUM_BATCH": "0"}, {"GEOAC_CHUNK_GIB": "1", "GEOAC_ACCUM_BATCH": "1"}]
    # the post-pass and the RK4 launches on disjoint sets of compute units (CU-masked streams; fans that are not hybrid)
    plans += [{"GEOAC_NO_PAIR": "1", "GEOAC_CU_SPLIT": "64"}]
    if eqname == "EQ_GLOBAL" and G.has_ab_kernels():              # (A/B builds only: `make AB=1`, GEOAC_LIB=<that build>)
        plans += [{"GEOAC_DUO": "1"}, {"GEOAC_DUO": "1", "GEOAC_COMPACT": "0"}, {"GEOAC_DUO": "1", "GEOAC_TWO_CHUNKS": "1", "GEOAC_S_ROWS": "3000"}]
        # TRIO=1: the three-wave kernel of round 4 (k_rk4_trio: the ray on one wave, ONE launch-angle system on each of two more, a ring of two message slots) on the
        # share of the fan that otherwise takes two lanes per ray - the shallow tenth, all of it, and with short epochs
        plans += [{"GEOAC_TRIO": "1"}, {"GEOAC_TRIO": "1", "GEOAC_PAIR_FRAC": "1.0"}, {"GEOAC_TRIO": "1", "GEOAC_TWO_CHUNKS": "1", "GEOAC_S_ROWS": "3000"}]
    for env in plans:
        rec, st = run(env)
        assert st == steps, env
        assert np.array_equal(rec, ref), env
