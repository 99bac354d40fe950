cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
python -m pytest tests/test_gpu_fullfan.py tests/test_gpu_eigenray.py tests/test_gpu_edges.py tests/test_gpu_cli.py tests/test_gpu_known_answers.py -m gpu -q > gpurun_out/r04j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04j/pytest.log
tail -8 gpurun_out/r04j/pytest.log
mkdir -p /tmp/wr1 /tmp/wr16 && cp tests/golden/ToyAtmo.met /tmp/wr1/ && cp tests/golden/ToyAtmo.met /tmp/wr16/
( cd /tmp/wr1 && time ( $GRAFT_REPO_ROOT/geoac_amd/bin/GeoAcGlobal -prop ToyAtmo.met phi_min=-180 phi_max=178 phi_step=2 gpu_fmt_threads=1 gpu_stats=stats.json > /dev/null ) ; cat stats.json ) > gpurun_out/r04j/writerays_1thread.log 2>&1
( cd /tmp/wr16 && time ( $GRAFT_REPO_ROOT/geoac_amd/bin/GeoAcGlobal -prop ToyAtmo.met phi_min=-180 phi_max=178 phi_step=2 gpu_stats=stats.json > /dev/null ) ; cat stats.json ; nproc ) > gpurun_out/r04j/writerays_default.log 2>&1
( cmp /tmp/wr1/ToyAtmo_raypaths.dat /tmp/wr16/ToyAtmo_raypaths.dat && cmp /tmp/wr1/ToyAtmo_results.dat /tmp/wr16/ToyAtmo_results.dat && echo "files identical"; md5sum /tmp/wr16/*.dat ) >> gpurun_out/r04j/writerays_default.log 2>&1
cat gpurun_out/r04j/writerays_1thread.log gpurun_out/r04j/writerays_default.log
