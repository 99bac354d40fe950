"""GPU test of the drop-in surface: the GeoAc2D / GeoAc3D / GeoAcGlobal / GeoAc3D.RngDep / GeoAcGlobal.RngDep -prop drivers of this repo (GPU fan behind
the C ABI) must write the same files as the reference's own binaries: same file set, same line structure, every
number equal to the printed precision (6-8 significant digits) up to one unit in the last printed place."""
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


@pytest.mark.parametrize("case", ["global", "3d", "2d", "global_norays", "3drd", "globalrd"])
def test_cli_files_match_reference_binaries(case, tmp_path):
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
    subprocess.run([exe, "-prop"] + inputs + params, cwd=tmp_path, check=True, stdout=subprocess.DEVNULL)
    want_files = sorted(f for f in os.listdir(gold) if f.endswith(".dat"))
    got_files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".dat") and not f.startswith("loc_"))
    assert got_files == want_files
    same = tot = 0
    for f in want_files:
        s, t = _compare_files(tmp_path / f, os.path.join(gold, f))
        same += s; tot += t
    print(f"{case}: {same}/{tot} tokens textually identical")
    assert same >= 0.995 * tot
