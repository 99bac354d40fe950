#!/bin/bash
# rocprofv3 evidence for BASELINE configs 2-5 (on the GPU box): every configuration's bench line (tools/bench_configs.py), kernel-trace stats of
# cfg2 / cfg3 / cfg4 / cfg5, one SQ PMC pass each for cfg2 / cfg3 / cfg5 and the six passes of tools/pmc_cfg.sh for cfg4, the whole 64-receiver
# ring of config 5 on one GPU (tools/perf_eigenray.py).  Everything lands in gpurun_out/<tag>/; the summaries to keep are copied to profiles/.
R=$GRAFT_REPO_ROOT; TAG=${1:-r03_cfg}; O=$R/gpurun_out/$TAG; mkdir -p $O
# (one process per configuration: the eigenray share measured behind the 1 M-ray fan in the same process reads 40 % high - allocator state)
cd $R && : > $O/configs.jsonl && for c in cfg1 cfg2 cfg3 cfg4_350 cfg4 cfg4_full cfg5; do timeout -k 10 300 python3 tools/bench_configs.py $c >> $O/configs.jsonl 2>> $O/configs.err; done
echo "configs done"
cd /tmp && export TMPDIR=/tmp
for c in cfg2 cfg3 cfg4 cfg5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_$c -o $c --output-format csv -- python3 $R/tools/bench_configs.py $c > $O/${c}_under_rocprof.json 2> $O/stats_$c.err || echo "stats $c failed"
  f=$(find $O/stats_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${c}_kernel_stats.csv
  echo "stats $c done"
done
for c in cfg2 cfg3 cfg5; do
  mkdir -p $O/pmc_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $O/pmc_$c --output-format csv -- python3 $R/tools/bench_configs.py $c > $O/pmc_$c.log 2>&1 || echo "pmc $c failed"
  (cd $R && python3 tools/pmc_summary.py $O/pmc_${c}_summary.json $O/pmc_$c > $O/pmc_${c}_summary.txt 2>&1) || true
  echo "pmc $c done"
done
cd $R && bash tools/pmc_cfg.sh cfg4 $TAG/pmc_cfg4 > $O/pmc_cfg4.txt 2>&1 || true
echo "pmc cfg4 done"
cd $R && (timeout -k 10 200 python3 tools/perf_eigenray.py global; timeout -k 10 300 python3 tools/perf_eigenray.py globalrd) > $O/ring64.txt 2>&1 || true
rm -f $O/midfans.txt; timeout -k 10 400 bash tools/sweep_midfans.sh $O/midfans.txt > /dev/null 2>&1 || true
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +1M -delete
cut -c1-300 $O/configs.jsonl; tail -5 $O/ring64.txt
