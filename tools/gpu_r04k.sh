cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04k
python tools/sweep_hybrid.py 6 global 0.10:0.75 0.08:0.75 0.06:0.75 0.12:0.75 0.10:0.70 0.10:0.72 0.10:0.78 0.10:0.82 0.08:0.72 0.10:0.75 > gpurun_out/r04k/sweep_global.log 2>&1; cat gpurun_out/r04k/sweep_global.log
python tools/sweep_hybrid.py 5 3d 0.25:0.80 0.20:0.80 0.30:0.80 0.25:0.75 0.25:0.85 0.25:0.80 > gpurun_out/r04k/sweep_3d.log 2>&1; cat gpurun_out/r04k/sweep_3d.log
for rows in 6144 8192 10240 12288; do GEOAC_DEBUG_ENV=1 GEOAC_S_ROWS=$rows GEOAC_AB_TAG="s_rows=$rows" python tools/ab_metric.py 5 geoac_amd/libgeoac_hip.so 2>&1 | tail -1; done > gpurun_out/r04k/rows.log 2>&1; cat gpurun_out/r04k/rows.log
