cd $GRAFT_REPO_ROOT
(timeout -k 10 300 python tools/bench_configs.py cfg4 > gpurun_out/two_a.log 2>&1; echo "A rc=$?" >> gpurun_out/two_a.log) &
(timeout -k 10 300 python tools/bench_configs.py cfg4 > gpurun_out/two_b.log 2>&1; echo "B rc=$?" >> gpurun_out/two_b.log) &
wait
tail -c 600 gpurun_out/two_a.log; echo; tail -c 600 gpurun_out/two_b.log
