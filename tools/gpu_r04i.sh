cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
python -m pytest tests -m gpu -q > gpurun_out/r04i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04i/pytest.log
tail -6 gpurun_out/r04i/pytest.log
bash tools/profile_round.sh r04_c > gpurun_out/r04i/profile_round.log 2>&1; tail -15 gpurun_out/r04i/profile_round.log
