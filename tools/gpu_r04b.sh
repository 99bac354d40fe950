set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
python -m pytest tests -m gpu -q > gpurun_out/r04b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04b/pytest.log
tail -15 gpurun_out/r04b/pytest.log
GEOAC_AB_SET=cfg3 python tools/ab_metric.py 3 build_ab_r03/libgeoac_hip.so geoac_amd/libgeoac_hip.so > gpurun_out/r04b/ab_cfg3.log 2>&1; cat gpurun_out/r04b/ab_cfg3.log
GEOAC_DEBUG_ENV=1 GEOAC_PP_LDS_TABLE=0 GEOAC_AB_SET=cfg3 python tools/ab_metric.py 3 geoac_amd/libgeoac_hip.so > gpurun_out/r04b/ab_cfg3_reg.log 2>&1; cat gpurun_out/r04b/ab_cfg3_reg.log
python tools/ab_metric.py 6 build_ab_r03/libgeoac_hip.so geoac_amd/libgeoac_hip.so build_ab_x15/libgeoac_hip.so > gpurun_out/r04b/ab_metric.log 2>&1; cat gpurun_out/r04b/ab_metric.log
mkdir -p /tmp/wr1 /tmp/wr16 && cp tests/golden/ToyAtmo.met /tmp/wr1/ && cp tests/golden/ToyAtmo.met /tmp/wr16/
( cd /tmp/wr1 && /usr/bin/time -v $GRAFT_REPO_ROOT/geoac_amd/bin/GeoAcGlobal -prop ToyAtmo.met phi_min=-180 phi_max=178 phi_step=2 gpu_fmt_threads=1 gpu_stats=stats.json > /dev/null 2> time.log; tail -20 time.log | grep -E "Elapsed|Maximum resident"; cat stats.json ) > gpurun_out/r04b/writerays_1thread.log 2>&1
( cd /tmp/wr16 && /usr/bin/time -v $GRAFT_REPO_ROOT/geoac_amd/bin/GeoAcGlobal -prop ToyAtmo.met phi_min=-180 phi_max=178 phi_step=2 gpu_stats=stats.json > /dev/null 2> time.log; tail -20 time.log | grep -E "Elapsed|Maximum resident"; cat stats.json ) > gpurun_out/r04b/writerays_default.log 2>&1
cmp /tmp/wr1/ToyAtmo_raypaths.dat /tmp/wr16/ToyAtmo_raypaths.dat && cmp /tmp/wr1/ToyAtmo_results.dat /tmp/wr16/ToyAtmo_results.dat && echo "files identical" >> gpurun_out/r04b/writerays_default.log
ls -la /tmp/wr16/ >> gpurun_out/r04b/writerays_default.log
cat gpurun_out/r04b/writerays_1thread.log gpurun_out/r04b/writerays_default.log
