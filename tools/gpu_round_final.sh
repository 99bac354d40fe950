cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round_final
python -m pytest tests -m gpu -q > gpurun_out/round_final/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/round_final/pytest.log
tail -4 gpurun_out/round_final/pytest.log
bash tools/profile_round.sh r04_c > gpurun_out/round_final/profile_round.log 2>&1; tail -12 gpurun_out/round_final/profile_round.log | cut -c1-400
python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 > gpurun_out/r04_c/bench_n2_gloo_rehearsal.json 2> gpurun_out/r04_c/bench_n2.err; python -c "
import json; d=json.load(open('gpurun_out/r04_c/bench_n2_gloo_rehearsal.json')); print(d['n_gpus'], d['value'], d['ms_per_step'], d.get('launcher'), 'cpu' in str(d.get('cpu_baseline'))[:5] or d.get('cpu_baseline',{}).get('value'))"
