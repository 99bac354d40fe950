"""GPU test of the drop-in surface: the GeoAc2D / GeoAc3D / GeoAcGlobal / GeoAc3D.RngDep / GeoAcGlobal.RngDep -prop drivers of this repo (GPU fan behind
the C ABI) must write the same files as the reference's own binaries: same file set, same line structure, every
number equal to the printed precision (6-8 significant digits) up to one unit in the last printed place."""
import json
import os
import shutil
import subprocess

import pytest

import harness as H

pytestmark = pytest.mark.gpu

CLI_GOLD = os.path.join(H.GOLDEN_DIR, "cli")
BIN = os.path.join(H.ROOT, "geoac_amd", "bin")


def _tokens_close(a, b):
    if a == b:
        return True
    try:
        x, y = float(a), float(b)
    except ValueError:
        return False
    # one unit in the last printed place of a 6-significant-digit number is <= 1e-5 relative
    return abs(x - y) <= 1.2e-5 * max(abs(x), abs(y)) + 1e-300


def _compare_files(got, want):
    gl, wl = open(got).read().split("\n"), open(want).read().split("\n")
    assert len(gl) == len(wl), f"{os.path.basename(want)}: {len(gl)} lines vs {len(wl)}"
    ntok = nsame = 0
    for i, (g, w) in enumerate(zip(gl, wl)):
        gt, wt = g.split("\t"), w.split("\t")
        assert len(gt) == len(wt), f"{os.path.basename(want)} line {i + 1}: column count"
        for a, b in zip(gt, wt):
            ntok += 1
            nsame += (a == b)
            assert _tokens_close(a, b), f"{os.path.basename(want)} line {i + 1}: {a!r} vs {b!r}"
    return nsame, ntok


@pytest.mark.parametrize("case", ["global", "3d", "2d", "cfg1", "global_norays", "3drd", "globalrd", "global+groups", "3drd+groups", "3drd+sub", "global+1thread", "globalrd+1thread"])
def test_cli_files_match_reference_binaries(case, tmp_path):
    """"+groups": the same fan integrated one azimuth group at a time (the path large WriteRays fans take: bounded sample list, text of
    group g written while group g+1 is on the GPU) must give the same files.  "+sub": the launch plan of saturated grid fans forced on the
    small one - one lane per ray, cooperative LDS-DMA gather, 512-row epochs in four sub-epochs handed from workgroup to workgroup - with the
    raypath samples and caustics on"""
    env = dict(os.environ)
    gpu_args = []                  # arguments of the GPU build (geoac_cli.cpp: gpu_rays_per_batch=, gpu_opt=KEY:value -> geoac_set_option); the environment is not read
    if case.endswith("+groups"):
        case = case[:-len("+groups")]
        gpu_args = ["gpu_rays_per_batch=2"]
    if case.endswith("+1thread"):                 # (the text formatted by one thread instead of the host's cores: the same bytes, test below)
        case = case[:-len("+1thread")]
        gpu_args = ["gpu_fmt_threads=1"]
    if case.endswith("+sub"):
        case = case[:-len("+sub")]
        gpu_args = ["gpu_opt=GRID_LANES:1", "gpu_opt=SPREAD:1", "gpu_opt=SUB_MIN_WAVES:0", "gpu_opt=SUB_EPOCHS:4", "gpu_opt=S_ROWS:512"]
    gold = os.path.join(CLI_GOLD, case)
    args = open(os.path.join(gold, "ARGS")).read().split()
    binary, params = args[0], args[1:]
    exe = os.path.join(BIN, binary)
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    if binary == "GeoAcGlobal.RngDep":
        import rngdep_data as RD
        RD.write_grid_global(str(tmp_path), short_paths=False)
        inputs = ["g", "loc_lat.dat", "loc_lon.dat"]
    elif binary.endswith("RngDep"):
        import rngdep_data as RD
        RD.write_grid(str(tmp_path), short_paths=False)      # the driver is run with relative names from cwd
        inputs = ["p", "loc_x.dat", "loc_y.dat"]
    else:
        shutil.copy(H.TOYATMO, tmp_path / "ToyAtmo.met")
        inputs = ["ToyAtmo.met"]
    subprocess.run([exe, "-prop"] + inputs + params + gpu_args, cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, env=env)
    want_files = sorted(f for f in os.listdir(gold) if f.endswith(".dat"))
    got_files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".dat") and not f.startswith("loc_"))
    assert got_files == want_files
    same = tot = 0
    for f in want_files:
        s, t = _compare_files(tmp_path / f, os.path.join(gold, f))
        same += s; tot += t
    print(f"{case}: {same}/{tot} tokens textually identical")
    assert same >= 0.995 * tot


@pytest.mark.parametrize("binary", ["GeoAcGlobal", "GeoAc3D"])
def test_files_do_not_depend_on_the_formatter_threads_or_the_groups(binary, tmp_path):
    """the -prop text is formatted in chunks of 64 rays on several threads and written in order; a chunk formats its first row per stream both ways -
    before and after the stream's sticky setprecision(8) (Q14) - and the writer picks.  A 12 x 23 fan with raypaths and caustics (five chunks, rays that
    break without a results row, legs without caustics) must give the SAME BYTES with 1, 3 and 16 threads, whole or in azimuth groups of 40 rays."""
    exe = os.path.join(BIN, binary)
    args = ["-prop", "ToyAtmo.met", "theta_min=1", "theta_max=45", "theta_step=2", "phi_min=-180", "phi_max=150", "phi_step=30", "bounces=2", "WriteCaustics=True"]
    outs = {}
    for tag, extra in (("t1", ["gpu_fmt_threads=1"]), ("t3", ["gpu_fmt_threads=3"]), ("t16", ["gpu_fmt_threads=16"]), ("t3g", ["gpu_fmt_threads=3", "gpu_rays_per_batch=40"])):
        d = tmp_path / tag; d.mkdir()
        shutil.copy(H.TOYATMO, d / "ToyAtmo.met")
        r = subprocess.run([exe] + args + extra + ["gpu_stats=" + str(d / "stats.json")], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()
        outs[tag] = {f: open(d / f, "rb").read() for f in sorted(os.listdir(d)) if f.endswith(".dat")}
        outs[tag]["stdout"] = r.stdout
        st = json.load(open(d / "stats.json"))
        assert st["text_bytes"] == sum(len(v) for k, v in outs[tag].items() if k.endswith(".dat") and k != "atmo.dat") - _header_bytes(outs[tag]) and st["text_MB_per_s"] > 0
    names = set(outs["t1"])
    assert {"ToyAtmo_results.dat", "ToyAtmo_raypaths.dat", "ToyAtmo_caustics-path0.dat", "ToyAtmo_caustics-path2.dat"} <= names
    assert len(outs["t1"]["ToyAtmo_raypaths.dat"]) > 1000000
    for tag in ("t3", "t16", "t3g"):
        assert set(outs[tag]) == names
        for f in names:
            assert outs[tag][f] == outs["t1"][f], f"{f} differs between one formatter thread and {tag}"


def _header_bytes(files):
    """bytes of the header lines ('# ...') of the result / raypath / caustic files: written when the files are opened, not by the formatter"""
    n = 0
    for k, v in files.items():
        if k.endswith(".dat") and k != "atmo.dat" and v.startswith(b"#"):
            n += v.index(b"\n") + 1
    return n


_NUM = __import__("re").compile(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?")


def _compare_text(got, want, what):
    """prompts / summaries: the text between the numbers must be identical, the numbers equal to the printed precision"""
    assert _NUM.sub("#", got) == _NUM.sub("#", want), f"{what}: text differs\n--- got\n{got}\n--- want\n{want}"
    gn, wn = _NUM.findall(got), _NUM.findall(want)
    same = 0
    for a, b in zip(gn, wn):
        same += (a == b)
        assert _tokens_close(a, b), f"{what}: {a!r} vs {b!r}"
    return same, len(wn)


@pytest.mark.parametrize("case", ["ia_global", "ia_global_noamp", "ia_3d", "ia_2d", "ia_3drd", "ia_globalrd"])
def test_cli_interactive_matches_reference_binaries(case, tmp_path):
    """-interactive sessions (two rays each): stdout (prompts, arrival summaries) and the raypath.dat / caustics.dat left by the last ray
    against the reference binaries' (tests/golden/make_golden_cli.py: IA_CASES)"""
    gold = os.path.join(CLI_GOLD, case)
    args = open(os.path.join(gold, "ARGS")).read().split()
    binary, opt, params = args[0], args[1], args[2:]
    stdin = open(os.path.join(gold, "STDIN")).read()
    exe = os.path.join(BIN, binary)
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    if binary == "GeoAcGlobal.RngDep":
        import rngdep_data as RD
        RD.write_grid_global(str(tmp_path), short_paths=False)
        inputs = ["g", "loc_lat.dat", "loc_lon.dat"]
    elif binary.endswith("RngDep"):
        import rngdep_data as RD
        RD.write_grid(str(tmp_path), short_paths=False)
        inputs = ["p", "loc_x.dat", "loc_y.dat"]
    else:
        shutil.copy(H.TOYATMO, tmp_path / "ToyAtmo.met")
        inputs = ["ToyAtmo.met"]
    r = subprocess.run([exe, opt] + inputs + params, cwd=tmp_path, check=True, stdout=subprocess.PIPE, input=stdin.encode(), timeout=600)
    same, tot = _compare_text(r.stdout.decode(), open(os.path.join(gold, "LOG.txt")).read(), "stdout")
    for f in ("raypath.dat", "caustics.dat"):
        if not os.path.exists(os.path.join(gold, f)):
            assert not os.path.exists(tmp_path / f), f
            continue
        assert os.path.exists(tmp_path / f), f
        s, t = _compare_files(tmp_path / f, os.path.join(gold, f))
        same += s; tot += t
    print(f"{case}: {same}/{tot} numbers textually identical")
    assert same >= 0.99 * tot
