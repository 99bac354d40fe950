"""No kernel the launch plan can select may touch scratch memory (private-segment spills sit on the serial chain of every RK4 step): read
from the compiler's own resource report of the shipped build (geoac_amd/csrc/build/*.resource_usage.txt, written by the Makefile with
-Rpass-analysis=kernel-resource-usage; hipcc cross-compiles for gfx950 without a GPU).  Kernels that only exist for A/B runs or tests -
selected by a GEOAC_* knob, never by geoac_fan_launch on its own - are listed with the knob that reaches them."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "geoac_amd", "csrc", "build")

# instantiations no default launch plan reaches (the knob that does), and set-up kernels that run once per atmosphere / parameter change
NOT_ON_THE_LAUNCH_PLAN = [
    (r"k_rk4<Eq(3D|Global)RngDep<(true|false), 1, false, false>", "one lane per ray with per-lane gathers: GEOAC_GRID_COOP=0 / GEOAC_GRID_LANES=1 with thinning, and the plan's "
                                                                    "fallback for evaluation tables of 4 GiB and more (100 x 100 x 1400 grids), which the cooperative kernel's 32-bit record offsets do not reach"),
    (r"k_postpass<Eq(Global|3D|2D)<(true|false)> >", "exact Sutherland-Bass evaluation at every midpoint: GEOAC_ABS_TABLE=0, and the plan's fallback when a table does not serve a profile "
                                                     "(GEOAC_FAN_ABS_FALLBACK); the default is k_postpass_tab"),
    (r"k_atab_build|k_gb_|k_probe_", "set-up / probe kernels, not per launch"),
]


# ON the launch plan and still using scratch: none.  (Until round 3 the four-lane per-lane-gather kernels of the grid sets - fans of 4 097 - 16 384 rays - kept an
# 88-byte table there: a select over the four corner nodes with a lane-dependent corner; the node index is computed by arithmetic now.)
KNOWN_ON_THE_PLAN = []


def _rows():
    files = [os.path.join(BUILD, f) for f in ("geoac_kernels.hip.resource_usage.txt", "geoac_gridbuild.hip.resource_usage.txt")]
    if not all(os.path.exists(f) for f in files):
        subprocess.check_call(["make", "-s", "-j", "4", "-C", os.path.join(ROOT, "geoac_amd", "csrc"), "ARCH=gfx950"])
    rows, cur = [], None
    for f in files:
        for line in open(f):
            m = re.search(r"remark: .*?(Function Name|VGPRs Spill|ScratchSize \[bytes/lane\]): (\S+)", line)
            if not m:
                continue
            if m.group(1) == "Function Name":
                cur = {"name": m.group(2)}
                rows.append(cur)
            elif cur is not None:
                cur[m.group(1).split(" [")[0]] = int(m.group(2))
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(rows, names):
        r["name"] = n.replace("void ", "")
    return rows


def test_launch_plan_kernels_use_no_scratch():
    rows = _rows()
    # the shipped build holds launch-plan kernels only: the diagnostic ones (k_rk4_duo, the grid sets' two-lane kernels) are compiled by `make AB=1`
    assert not [r["name"] for r in rows if "k_rk4_duo" in r["name"] or "k_rk4_trio" in r["name"] or re.search(r"k_rk4<Eq(3D|Global)RngDep<(true|false), 2,", r["name"])]
    assert len(rows) > 80                                            # the whole family of instantiations was seen
    offenders, excused, known = [], 0, []
    for r in rows:
        if r.get("ScratchSize", 0) == 0:
            continue
        if any(re.search(pat, r["name"]) for pat, _ in NOT_ON_THE_LAUNCH_PLAN):
            excused += 1
            continue
        if any(re.search(pat, r["name"]) for pat in KNOWN_ON_THE_PLAN):
            known.append(r["name"])
            continue
        offenders.append(f'{r["name"]}: {r["ScratchSize"]} B/lane scratch, {r.get("VGPRs Spill", 0)} spilled VGPRs')
    print(f"{len(rows)} kernels, {excused} A/B-only or set-up kernels with scratch, {len(known)} known exceptions on the plan, "
          f"{len(offenders)} other launch-plan kernels with scratch")
    assert not known
    assert not offenders, "\n".join(offenders)
