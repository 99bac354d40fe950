#!/bin/bash
# rocprofv3 evidence for BASELINE configs 4 and 5 (on the GPU box): kernel-trace stats of tools/bench_configs.py cfg4 / cfg5 and the PMC
# passes of tools/pmc_cfg.sh for cfg4; everything lands in gpurun_out/<tag>/, the summaries to keep are copied to profiles/ afterwards.
set -e
R=$GRAFT_REPO_ROOT; TAG=${1:-r02_cfg}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R && python3 tools/bench_configs.py cfg1 cfg2 cfg3 cfg4_350 cfg4 cfg4_full cfg5 > $O/configs.jsonl 2> $O/configs.err
cd /tmp && export TMPDIR=/tmp
for c in cfg4 cfg5; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats_$c -o $c --output-format csv -- python3 $R/tools/bench_configs.py $c > $O/${c}_under_rocprof.json 2> $O/stats_$c.err || echo "stats $c failed"
  f=$(find $O/stats_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${c}_kernel_stats.csv
done
cd $R && bash tools/pmc_cfg.sh cfg4 $TAG/pmc_cfg4 > $O/pmc_cfg4.txt 2>&1 || true
find $O -name "*kernel_trace.csv" -size +2M -delete
cat $O/configs.jsonl | cut -c1-400; head -5 $O/cfg4_kernel_stats.csv
