cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
GEOAC_AB_SET=cfg3 python tools/ab_metric.py 3 build_ab_r03/libgeoac_hip.so geoac_amd/libgeoac_hip.so > gpurun_out/r04g/cfg3.log 2>&1; cat gpurun_out/r04g/cfg3.log
GEOAC_DEBUG_ENV=1 GEOAC_RK4_PREFETCH=0 GEOAC_AB_TAG="prefetch=0" GEOAC_AB_SET=cfg3 python tools/ab_metric.py 3 geoac_amd/libgeoac_hip.so >> gpurun_out/r04g/cfg3.log 2>&1; tail -2 gpurun_out/r04g/cfg3.log
python tools/ab_metric.py 6 build_ab_r03/libgeoac_hip.so geoac_amd/libgeoac_hip.so > gpurun_out/r04g/metric.log 2>&1; cat gpurun_out/r04g/metric.log
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fullfan.py -m gpu -q -x > gpurun_out/r04g/pytest.log 2>&1; tail -3 gpurun_out/r04g/pytest.log
