cd $GRAFT_REPO_ROOT
bash tools/profile_configs.sh r04_c_cfg > gpurun_out/r04_c_cfg.log 2>&1; tail -12 gpurun_out/r04_c_cfg.log | cut -c1-300
