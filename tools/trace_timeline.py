"""per-launch timeline from a rocprofv3 --kernel-trace CSV: start / end (ms from the first listed launch) of the last N kernel launches"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    name = r["Kernel_Name"]
    short = "pair" if "EqGlobalPair" in name else "rk4-1" if "k_rk4" in name else "post" if "k_postpass" in name else "accum" if "k_accum" in name else name[:12]
    print(f"{short:6s} {a:8.2f} -> {b:8.2f}  ({b - a:6.2f} ms)  grid {r.get('Grid_Size', '?')} wg {r.get('Workgroup_Size', '?')}")
