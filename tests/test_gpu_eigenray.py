"""GPU test of the eigenray modes (-eig_search / -eig_direct of the spherical mains): the batched search of this repo
(geoac_eig_search: one fan launch per decision round instead of one propagation per ray) must identify the reference's
eigenrays through the reference's iteration sequence - same verbose log line for line, same result and raypath files,
every number equal to its printed precision."""
import os
import shutil
import subprocess

import pytest

import harness as H
from test_gpu_cli import _compare_files, _tokens_close

pytestmark = pytest.mark.gpu

CLI_GOLD = os.path.join(H.GOLDEN_DIR, "cli")
BIN = os.path.join(H.ROOT, "geoac_amd", "bin")


from parity import compare_logs as _compare_logs  # noqa: E402


@pytest.mark.parametrize("case", ["eig_global", "eig_global_direct", "eig_globalrd", "eig_globalrd_direct", "eig_3d", "eig_3d_direct", "eig_3drd", "eig_3drd_direct"])
def test_eigenray_modes_match_reference_binaries(case, tmp_path):
    gold = os.path.join(CLI_GOLD, case)
    args = open(os.path.join(gold, "ARGS")).read().split()
    binary, opt, params = args[0], args[1], args[2:]
    exe = os.path.join(BIN, binary)
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    if binary == "GeoAcGlobal.RngDep":
        import rngdep_data as RD
        RD.write_grid_global(str(tmp_path), short_paths=False)
        inputs = ["g", "loc_lat.dat", "loc_lon.dat"]
    elif binary == "GeoAc3D.RngDep":
        import rngdep_data as RD
        RD.write_grid(str(tmp_path), short_paths=False)
        inputs = ["p", "loc_x.dat", "loc_y.dat"]
    else:
        shutil.copy(H.TOYATMO, tmp_path / "ToyAtmo.met")
        inputs = ["ToyAtmo.met"]
    r = subprocess.run([exe, opt] + inputs + params, cwd=tmp_path, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print(r.stderr.decode().strip())
    want_files = sorted(f for f in os.listdir(gold) if f.endswith(".dat"))
    got_files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".dat") and not f.startswith("loc_"))
    assert got_files == want_files
    _compare_logs(r.stdout.decode(), open(os.path.join(gold, "LOG.txt")).read())
    for f in want_files:
        if "Eigenray-" in f:
            # amplitude [dB] near a zero crossing and the tiny attenuation of the first rows: compare on an absolute dB scale too
            gl, wl = open(tmp_path / f).read().split("\n"), open(os.path.join(gold, f)).read().split("\n")
            assert len(gl) == len(wl)
            for i, (g, w) in enumerate(zip(gl, wl)):
                gt, wt = g.split("\t"), w.split("\t")
                assert len(gt) == len(wt)
                for a, b in zip(gt, wt):
                    if a == b:
                        continue
                    x, y = float(a), float(b)
                    assert abs(x - y) <= 1.2e-5 * max(abs(x), abs(y)) + 1e-5, f"{f} line {i + 1}: {a!r} vs {b!r}"
        else:
            gl, wl = open(tmp_path / f).read(), open(os.path.join(gold, f)).read()
            _compare_logs(gl, wl)


def _ring(lat0, lon0, n, radius_deg=2.5):
    """n receivers on a ring of `radius_deg` degrees of arc around (lat0, lon0) (config 5 of SURVEY §8d)"""
    import numpy as np
    az = np.arange(n) * (2.0 * np.pi / n)
    lat = lat0 + radius_deg * np.cos(az)
    lon = lon0 + radius_deg * np.sin(az) / np.cos(np.radians(lat0))
    return np.stack([lat, lon], axis=1)


def test_batched_receivers_equal_single_receiver_searches():
    """batching is invisible: the eigenrays found for a receiver inside a 16-receiver batch are bit-identical to the ones of
    a search run for that receiver alone, and the batch needs far fewer fan launches than the sum of the single searches"""
    import numpy as np
    import geoac_amd as G
    ctx = G.FanContext(G.EQ_GLOBAL, device=0)
    ctx.load_met(H.TOYATMO)
    ctx.set_params(src=(0.0, 30.0, 0.0))
    rcv = _ring(30.0, 0.0, 16)
    all_ = ctx.eig_search(rcv, bnc_min=0, bnc_max=0)
    assert len(all_["eig"]) >= 2                       # ToyAtmo: stratospheric arrivals on the downwind (western) side of the ring
    launches_single = 0
    with_rays = sorted(set(int(r) for r in all_["eig"][:, G.EIG["RCVR"]]))
    picks = sorted(set(with_rays[:3] + [0, 5]))
    for i in picks:
        one = ctx.eig_search(rcv[i:i + 1], bnc_min=0, bnc_max=0)
        launches_single += one["stats"]["launches"]
        sel = all_["eig"][all_["eig"][:, G.EIG["RCVR"]] == i]
        assert len(sel) == len(one["eig"])
        a = sel[:, 1:G.EIG["SMP0"]]; b = one["eig"][:, 1:G.EIG["SMP0"]]
        assert np.array_equal(a, b)
        for k in range(len(sel)):
            s0, n = int(sel[k, G.EIG["SMP0"]]), int(sel[k, G.EIG["NSMP"]])
            t0 = int(one["eig"][k, G.EIG["SMP0"]])
            assert np.array_equal(all_["smp"][s0:s0 + n, 1:], one["smp"][t0:t0 + n, 1:])
    print("16 receivers:", all_["stats"], ";", len(picks), "single searches:", launches_single, "launches; receivers with eigenrays:", with_rays)
    assert all_["stats"]["launches"] < 16 * launches_single / len(picks) / 2     # rounds mix scan and refinement groups
    # every eigenray really reaches its receiver: arrival within the refinement's 0.1 km tolerance is implied by identification;
    # celerity = distance / travel time must be acoustic
    assert ((all_["eig"][:, G.EIG["CELERITY"]] > 0.2) & (all_["eig"][:, G.EIG["CELERITY"]] < 0.36)).all()


def test_more_receivers_than_one_group_and_parallel_bounce_counts():
    """100 receivers (processed in groups of 96) with bounce counts 0..1 searched concurrently: every receiver's eigenrays and log are
    those of the same receiver searched alone"""
    import numpy as np
    import geoac_amd as G
    ctx = G.FanContext(G.EQ_GLOBAL, device=0)
    ctx.load_met(H.TOYATMO)
    ctx.set_params(src=(0.0, 30.0, 0.0))
    rcv = _ring(30.0, 0.0, 100)
    all_ = ctx.eig_search(rcv, bnc_min=0, bnc_max=1, verbose=True)
    with_rays = sorted(set(int(r) for r in all_["eig"][:, G.EIG["RCVR"]]))
    assert len(with_rays) >= 5
    for i in sorted(set(with_rays[:2] + [with_rays[-1], 97, 99])):
        one = ctx.eig_search(rcv[i:i + 1], bnc_min=0, bnc_max=1, verbose=True)
        sel = all_["eig"][all_["eig"][:, G.EIG["RCVR"]] == i]
        assert len(sel) == len(one["eig"])
        assert np.array_equal(sel[:, 1:G.EIG["SMP0"]], one["eig"][:, 1:G.EIG["SMP0"]])
        assert all_["logs"][i] == one["logs"][0]
        # eigenrays are numbered in the reference's order: by bounce count, then by inclination
        assert list(sel[:, G.EIG["INDEX"]]) == list(range(len(sel)))
        assert list(sel[:, G.EIG["BOUNCES"]]) == sorted(sel[:, G.EIG["BOUNCES"]])


@pytest.mark.parametrize("eqname,amp", [("EQ_GLOBAL", 1), ("EQ_GLOBAL", 0), ("EQ_3D", 1), ("EQ_3D", 0), ("EQ_3D_RNGDEP", 1), ("EQ_3D_RNGDEP", 0)])
def test_leg_records_do_not_depend_on_the_number_of_bounces(eqname, amp, tmp_path):
    """The eigenray scheduler integrates the inclination scans of one round that differ only in the bounce count ONCE, with the largest count, and hands the others their
    legs (geoac_eigenray.cpp, serve) - for all four equation sets that have searches.  That is only right if a ray launched with b bounces leaves, for its legs 0 .. a,
    bit for bit the records of the same ray launched with a bounces: pinned here for the stratified spherical and Cartesian sets (pair, one-lane and hybrid plans, table
    post-pass) and the Cartesian grid set, with and without amplitudes (tests/test_gpu_globalrd.py holds the same test for the spherical grid set)."""
    import numpy as np
    import geoac_amd as G
    eq = getattr(G, eqname)
    th = np.linspace(1.0, 40.0, 79); ph = np.full_like(th, -77.0)
    recs = {}
    for b in (0, 1, 2):
        ctx = G.FanContext(eq, device=0)
        if eqname == "EQ_3D_RNGDEP":
            import rngdep_data as RD
            ctx.load_grid(*RD.write_grid(str(tmp_path), short_paths=False))
            ctx.set_params(bounces=b, calc_amp=amp, mode=0, src=(0.0, 0.0, 0.0))
        else:
            ctx.load_met(H.TOYATMO)
            ctx.set_params(bounces=b, calc_amp=amp, mode=0)
        recs[b] = ctx.run(th, ph)[0].copy(); ctx.close()
    assert (recs[2][..., G.REC["VALID"]] > 0).sum() > 40
    for a in (0, 1):
        for b in range(a + 1, 3):
            assert np.array_equal(recs[b][:, :a + 1].view(np.uint64), recs[a].view(np.uint64)), (eqname, amp, a, b)


@pytest.mark.parametrize("binary", ["GeoAcGlobal", "GeoAc3D"])
def test_search_over_several_bounce_counts_does_not_depend_on_the_scan_merge(binary, tmp_path):
    """-eig_search with bnc_min < bnc_max on a STRATIFIED set: the scans of the three bounce counts are merged into one fan (serve); with the merge off
    (GEOAC_EIG_MERGE=0 under GEOAC_DEBUG_ENV=1: every request integrates its own rays) the log and the result files must be the same bytes."""
    exe = os.path.join(BIN, binary)
    if binary == "GeoAcGlobal":
        params = ["lat_src=30.0", "lon_src=0.0", "lat_rcvr=30.0", "lon_rcvr=-3.2", "bnc_min=0", "bnc_max=2", "verbose=True"]
    else:
        params = ["x_rcvr=-320.0", "y_rcvr=15.0", "bnc_min=0", "bnc_max=2", "verbose=True"]
    outs = {}
    for tag, env in (("merge", {}), ("nomerge", {"GEOAC_DEBUG_ENV": "1", "GEOAC_EIG_MERGE": "0"})):
        d = tmp_path / tag; d.mkdir()
        shutil.copy(H.TOYATMO, d / "ToyAtmo.met")
        r = subprocess.run([exe, "-eig_search", "ToyAtmo.met"] + params, cwd=d, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs[tag] = (r.stdout, {f: open(d / f, "rb").read() for f in sorted(os.listdir(d)) if f.endswith(".dat") and f != "atmo.dat"}, r.stderr.decode())
    assert outs["merge"][0] == outs["nomerge"][0], "verbose log differs with the scan merge off"
    assert outs["merge"][1] == outs["nomerge"][1] and len(outs["merge"][1]) >= 1
    assert len(outs["merge"][0]) > 2000
    # (the merge really happened: fewer rays integrated with it on)
    import re
    rays = {k: int(re.search(r"(\d+) rays in", v[2]).group(1)) for k, v in outs.items()}
    print(binary, "rays integrated with / without the merge:", rays)
    assert rays["merge"] < rays["nomerge"]
