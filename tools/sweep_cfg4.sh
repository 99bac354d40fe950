#!/bin/bash
export GEOAC_DEBUG_ENV=1      # A/B sweeps drive the launch-plan options through the environment (read only with this set)
# config-4 share under launch-plan knobs (sub-epochs, epoch length): one line each.  usage (GPU box): tools/sweep_cfg4.sh
cd $GRAFT_REPO_ROOT
for kv in "GEOAC_SUB_EPOCHS=1" "GEOAC_SUB_EPOCHS=4" "GEOAC_SUB_EPOCHS=8" "GEOAC_SUB_EPOCHS=16" "GEOAC_SUB_EPOCHS=8 GEOAC_S_ROWS=4096" "GEOAC_SUB_EPOCHS=4 GEOAC_S_ROWS=3072" "GEOAC_SUB_EPOCHS=16 GEOAC_S_ROWS=12288"; do
  r=$(env $kv timeout -k 10 120 python3 tools/bench_configs.py cfg4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f s  %.3e steps/s  rk4 %.0f ms  launches %d' % (d['seconds'], d['ray_steps_per_s'], d['roofline']['rk4_ms_per_pass'], d['roofline']['launches_per_pass']))")
  echo "$kv: $r"
done
