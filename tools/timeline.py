"""Kernel timeline of a rocprofv3 --kernel-trace csv: start / duration of every launch longer than `min_ms`, relative to the first one.
usage: timeline.py <kernel_trace.csv> [min_ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (e - s) / 1e6 < min_ms: continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    print(f"{(s - t0) / 1e6:10.2f} ms  +{(e - s) / 1e6:9.2f} ms  stream {r.get('Stream_Id', '?'):>3}  {name}")
