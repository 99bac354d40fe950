#!/bin/bash
# rocprofv3 kernel-trace + stats of one python command; summary (kernel, calls, total ms, avg ms) under gpurun_out/<tag>/.  usage: prof_stats.sh TAG script.py [args...]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o $TAG --output-format csv -- python3 $R/$1 "${@:2}" > $O/run.out 2> $O/run.err
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $O/kernel_stats_summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print(f"{r['Name'][:110]:<110s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:10.2f} avg_ms {float(r['AverageNs'])/1e6:9.3f} pct {r['Percentage']}")
PY
find $O/stats -name "*kernel_trace.csv" -size +6M -delete
cat $O/kernel_stats_summary.txt | head -14
