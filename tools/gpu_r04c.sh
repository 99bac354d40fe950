cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
python -m pytest tests -m gpu -q > gpurun_out/r04c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04c/pytest.log
tail -12 gpurun_out/r04c/pytest.log
export GEOAC_DEBUG_ENV=1 GEOAC_AB_SET=cfg3
for cfg in "0 8192" "0 61440" "0 100000" "1 38912" "1 61440" "1 100000"; do
  set -- $cfg
  GEOAC_AB_TAG="tbl=$1 pad=$2" GEOAC_PP_LDS_TABLE=$1 GEOAC_PP_LDS_PAD=$2 python tools/ab_metric.py 3 geoac_amd/libgeoac_hip.so 2>&1 | tail -2
done > gpurun_out/r04c/cfg3_occupancy.log 2>&1
GEOAC_AB_TAG="r03" python tools/ab_metric.py 3 build_ab_r03/libgeoac_hip.so >> gpurun_out/r04c/cfg3_occupancy.log 2>&1
cat gpurun_out/r04c/cfg3_occupancy.log
unset GEOAC_DEBUG_ENV GEOAC_AB_SET
python tools/ab_metric.py 6 build_ab_r03/libgeoac_hip.so geoac_amd/libgeoac_hip.so > gpurun_out/r04c/ab_metric.log 2>&1; cat gpurun_out/r04c/ab_metric.log
mkdir -p /tmp/wr1 /tmp/wr16 && cp tests/golden/ToyAtmo.met /tmp/wr1/ && cp tests/golden/ToyAtmo.met /tmp/wr16/
( cd /tmp/wr1 && time ( $GRAFT_REPO_ROOT/geoac_amd/bin/GeoAcGlobal -prop ToyAtmo.met phi_min=-180 phi_max=178 phi_step=2 gpu_fmt_threads=1 gpu_stats=stats.json > /dev/null ) ; cat stats.json ) > gpurun_out/r04c/writerays_1thread.log 2>&1
( cd /tmp/wr16 && time ( $GRAFT_REPO_ROOT/geoac_amd/bin/GeoAcGlobal -prop ToyAtmo.met phi_min=-180 phi_max=178 phi_step=2 gpu_stats=stats.json > /dev/null ) ; cat stats.json ; nproc ) > gpurun_out/r04c/writerays_default.log 2>&1
( cmp /tmp/wr1/ToyAtmo_raypaths.dat /tmp/wr16/ToyAtmo_raypaths.dat && cmp /tmp/wr1/ToyAtmo_results.dat /tmp/wr16/ToyAtmo_results.dat && echo "files identical"; ls -la /tmp/wr16/ ) >> gpurun_out/r04c/writerays_default.log 2>&1
cat gpurun_out/r04c/writerays_1thread.log gpurun_out/r04c/writerays_default.log
python bench.py --steps 5 --warmup 2 > gpurun_out/r04c/bench.json 2> gpurun_out/r04c/bench.err; tail -c 3000 gpurun_out/r04c/bench.json; tail -5 gpurun_out/r04c/bench.err
