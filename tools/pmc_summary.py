"""Aggregate rocprofv3 --pmc CSV output per kernel.  usage: pmc_summary.py <out.json> <dir_or_csv> [<dir_or_csv> ...]
Each argument is the output directory of ONE `rocprofv3 --kernel-trace --pmc <counters> --output-format csv` pass (counters are
collected in separate passes, never together with other trace domains).  Sums every counter per kernel name and records the call
count and the average kernel duration under that pass.  FETCH_SIZE / WRITE_SIZE come in KiB and are converted to bytes (no x2)."""
import csv, glob, json, os, sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    i = name.find("(")
    return name[:i] if i > 0 else name


def main():
    out, srcs = sys.argv[1], sys.argv[2:]
    K = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(set))
    dur = defaultdict(lambda: defaultdict(dict))
    for src in srcs:
        files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"]); c = row["Counter_Name"]; v = float(row["Counter_Value"])
                if c in ("FETCH_SIZE", "WRITE_SIZE"):
                    v *= 1024.0
                K[k][c] += v
                calls[k][c].add(row["Dispatch_Id"])
                if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    dur[k][c][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
    res = {}
    for k in K:
        res[k] = {}
        for c in K[k]:
            n = len(calls[k][c])
            res[k][c] = {"total": K[k][c], "calls": n, "per_call": K[k][c] / max(n, 1)}
            if dur[k][c]:
                res[k][c]["avg_ms"] = sum(dur[k][c].values()) / len(dur[k][c])
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k in sorted(res):
        print(k, {c: f"{v['total']:.4g}" for c, v in res[k].items()})


if __name__ == "__main__":
    main()
