# the A/B-only kernels (k_rk4_duo, the grid sets' two-lane kernels) under the schedule-independence and variant tests: an AB build (tools/build_ab.sh full AB=1) through GEOAC_LIB
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round_ab
GEOAC_LIB=$GRAFT_REPO_ROOT/build_ab_full/libgeoac_hip.so python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_gridbuild.py tests/test_gpu_rngdep.py tests/test_gpu_globalrd.py -m gpu -q > gpurun_out/round_ab/pytest_ab.log 2>&1; echo "rc=$?" >> gpurun_out/round_ab/pytest_ab.log
tail -5 gpurun_out/round_ab/pytest_ab.log
