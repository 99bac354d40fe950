"""Compiles geoac_kernels.hip / geoac_gridbuild.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and prints one line per
kernel: registers, spills, scratch, occupancy.  usage: resource_usage.py [substring filter ...] [--fail-on-scratch PATTERN]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "geoac_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [o.replace("(GeoacDevParams)", "").replace("void ", "") for o in out]


def table(src):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", "-o", "/dev/null", "-x", "hip",
           os.path.join(CSRC, src), "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key.split(" [")[0]] = val
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["name"] = n
    return rows


def main():
    args = sys.argv[1:]
    fail_pat = None
    if "--fail-on-scratch" in args:
        i = args.index("--fail-on-scratch"); fail_pat = args[i + 1]; del args[i:i + 2]
    bad = 0
    for src in ("geoac_kernels.hip", "geoac_gridbuild.hip"):
        for r in table(src):
            if args and not any(a in r["name"] for a in args):
                continue
            print(f'{r["name"]:<70s} VGPR {r.get("VGPRs", "?"):>3s} AGPR {r.get("AGPRs", "?"):>3s} spillV {r.get("VGPRs Spill", "?"):>3s} spillS {r.get("SGPRs Spill", "?"):>3s} '
                  f'scratch {r.get("ScratchSize", "?"):>4s} occ {r.get("Occupancy", "?")}')
            if fail_pat and re.search(fail_pat, r["name"]) and int(r.get("ScratchSize", "0")) > 0:
                bad += 1
    if bad:
        sys.exit(f"{bad} kernels matching {fail_pat!r} use scratch")


if __name__ == "__main__":
    main()
