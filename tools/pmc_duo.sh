#!/bin/bash
# PMC passes over the metric fan (bench.py, one pass): SQ issue / wait / LDS counters per kernel.  usage: pmc_duo.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-pmc_duo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1); mkdir -p $O/$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $O/$tag --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/$tag.json 2> $O/$tag.err || { tail -3 $O/$tag.err; echo "pass $tag failed"; }
done
cd $R && python3 tools/pmc_summary.py $O/summary.json $O/SQ_WAVE_CYCLES $O/SQ_INSTS_VALU > $O/summary.txt 2>&1
find $O -name "*counter_collection.csv" -size +1M -delete
grep "k_rk4\|postpass_tab" $O/summary.txt | cut -c1-900
