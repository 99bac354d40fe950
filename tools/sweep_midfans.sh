#!/bin/bash
export GEOAC_DEBUG_ENV=1      # A/B sweeps drive the launch-plan options through the environment (read only with this set)
# mid-size grid fans (4 097 - 32 768 rays) under the candidate launch plans: the multi-lane kernels without the record cache (the round-2
# default there) against the cooperative one-lane kernel.  usage: sweep_midfans.sh OUTFILE
R=${GRAFT_REPO_ROOT:-/root/repo}; O=${1:-$R/gpurun_out/midfans.txt}
for spec in "200 40 3d 1" "600 40 3d 1" "200 40 global 4" "600 40 global 4"; do
  for env in "" "GEOAC_GRID_LANES=1 GEOAC_SPREAD=1" "GEOAC_GRID_LANES=4 GEOAC_QUAD_CACHE=0" "GEOAC_GRID_LANES=2"; do
    echo "== $spec | ${env:-default}" >> $O
    env $env timeout -k 10 200 python3 $R/tools/perf_rngdep.py $spec 2>&1 | grep "CalcAmp" >> $O
  done
done
cat $O
