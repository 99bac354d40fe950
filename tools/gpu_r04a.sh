set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
python -m pytest tests -m gpu -x -q > gpurun_out/r04a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04a/pytest.log
tail -5 gpurun_out/r04a/pytest.log
python tools/amp_loose.py > gpurun_out/r04a/amp_loose.log 2>&1; tail -3 gpurun_out/r04a/amp_loose.log
python tools/ab_metric.py 6 geoac_amd/libgeoac_hip.so build_ab_x3/libgeoac_hip.so build_ab_x15/libgeoac_hip.so > gpurun_out/r04a/ab_metric.log 2>&1; cat gpurun_out/r04a/ab_metric.log
GEOAC_AB_SET=cfg3 python tools/ab_metric.py 3 geoac_amd/libgeoac_hip.so > gpurun_out/r04a/ab_cfg3.log 2>&1; cat gpurun_out/r04a/ab_cfg3.log
GEOAC_DEBUG_ENV=1 GEOAC_PP_LDS_TABLE=0 GEOAC_AB_SET=cfg3 python tools/ab_metric.py 3 geoac_amd/libgeoac_hip.so > gpurun_out/r04a/ab_cfg3_reg.log 2>&1; cat gpurun_out/r04a/ab_cfg3_reg.log
python tools/perf_sparse_waves.py > gpurun_out/r04a/sparse.log 2>&1; cat gpurun_out/r04a/sparse.log
