set -e
R=$GRAFT_REPO_ROOT; TAG=${1:-r1x}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R && timeout -k 10 400 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o $TAG --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/stats.err
for set in "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1); mkdir -p $O/pmc/$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc/$tag --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/pmc/$tag.json 2> $O/pmc/$tag.err
  echo "pmc pass $tag done"
done
cd $R && python3 tools/pmc_summary.py $O/pmc_summary.json $O/pmc/SQ_INSTS_VALU_ADD_F64 $O/pmc/SQ_INSTS_VALU $O/pmc/FETCH_SIZE $O/pmc/WRITE_SIZE > $O/pmc_summary.txt
steps=$(python3 -c "import json;print(json.load(open('$O/pmc/FETCH_SIZE.json'))['config']['ray_steps_per_pass'])")
python3 tools/pmc_derive.py $O/pmc_summary.json $steps $O/pmc_traffic.json $R/geoac_amd/libgeoac_hip.so > /dev/null
find $O/pmc -name "*counter_collection.csv" -size +1M -delete; find $O -name "*kernel_trace.csv" -size +4M -delete
cat $O/bench.json; cat $O/bench_under_rocprof.json; grep -E "flop_per|bytes_per|valu" $O/pmc_traffic.json; ls $O/stats/* | head
