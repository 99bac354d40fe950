set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_grid; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  tag=$(echo $set | cut -d' ' -f1); mkdir -p $O/$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set -d $O/$tag --output-format csv -- python3 $R/tools/bench_configs.py cfg4 > $O/$tag.log 2>&1 || { tail -5 $O/$tag.log; echo "pass $tag failed"; }
done
cd $R && python3 tools/pmc_summary.py $O/summary.json $O/SQ_WAVE_CYCLES $O/SQ_INSTS_VALU > $O/summary.txt 2>&1 || true
python3 - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_grid/summary.json"))
for k in d:
    if "k_rk4" in k or "postpass" in k:
        print(k[:70])
        for c,v in sorted(d[k].items()): print("   %-24s %.4g  (calls %d, avg %.3f ms)"%(c,v["total"],v["calls"],v.get("avg_ms",0)))
PY
find $O -name "*counter_collection.csv" -size +1M -delete
