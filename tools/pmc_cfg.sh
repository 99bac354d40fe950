#!/bin/bash
# PMC passes over one configuration of tools/bench_configs.py (separate rocprofv3 --pmc runs, kernel trace only), summary of the k_rk4 /
# k_postpass kernels in gpurun_out/<tag>/summary.txt.   usage (on the GPU box): tools/pmc_cfg.sh <cfg> <tag> [ENV=VAL ...]
set -e
CFG=$1; TAG=$2; shift 2
for kv in "$@"; do export "$kv"; done
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64"; do
  i=$((i+1)); mkdir -p $O/p$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $O/p$i --output-format csv -- python3 $R/tools/bench_configs.py $CFG > $O/p$i.log 2>&1 || { tail -3 $O/p$i.log; echo "pass $i ($set) failed"; }
done
cd $R && python3 tools/pmc_summary.py $O/summary.json $O/p* > /dev/null 2>&1 || true
python3 - <<PY
import json
d=json.load(open("$O/summary.json"))
with open("$O/summary.txt","w") as fh:
    for k in sorted(d):
        if "k_rk4" in k or "postpass" in k:
            fh.write(k[:90]+"\n")
            for c,v in sorted(d[k].items()): fh.write("   %-34s %.5g  (calls %d, avg %.3f ms)\n"%(c,v["total"],v["calls"],v.get("avg_ms",0)))
print(open("$O/summary.txt").read())
PY
find $O -name "*counter_collection.csv" -size +1M -delete; find $O -name "*.csv" -size +4M -delete
