cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
python -m pytest tests -m gpu -q -x > gpurun_out/r04f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04f/pytest.log
tail -8 gpurun_out/r04f/pytest.log
GEOAC_AB_SET=cfg3 python tools/ab_metric.py 3 build_ab_r03/libgeoac_hip.so geoac_amd/libgeoac_hip.so > gpurun_out/r04f/cfg3.log 2>&1; cat gpurun_out/r04f/cfg3.log
python tools/ab_metric.py 6 build_ab_r03/libgeoac_hip.so geoac_amd/libgeoac_hip.so > gpurun_out/r04f/metric.log 2>&1; cat gpurun_out/r04f/metric.log
python bench.py --steps 5 --warmup 2 > gpurun_out/r04f/bench.json 2> gpurun_out/r04f/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r04f/bench.json')); print(d['value'], d['ms_per_step'], d['roofline'].get('issue')); print(json.dumps(d['other_scalings'])[:3000])"; tail -3 gpurun_out/r04f/bench.err
python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 > gpurun_out/r04f/bench_n2.json 2> gpurun_out/r04f/bench_n2.err; python -c "
import json; d=json.load(open('gpurun_out/r04f/bench_n2.json')); print(d['n_gpus'], d['value'], d['ms_per_step'], d.get('launcher')); print(json.dumps(d['other_scalings'])[:3000])"; tail -3 gpurun_out/r04f/bench_n2.err
