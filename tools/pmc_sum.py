"""sums of rocprofv3 --pmc counters per kernel: usage pmc_sum.py <dir with *counter_collection.csv> -> table kernel x counter (totals over the dispatches)"""
import csv, glob, sys, collections, re
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(tot, key=lambda k: -sum(tot[k].values())):
    print(k, {c: "%.4g" % v for c, v in sorted(tot[k].items())}, "dispatches", max(n[(k, c)] for c in tot[k]))
