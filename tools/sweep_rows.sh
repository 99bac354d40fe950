#!/bin/bash
export GEOAC_DEBUG_ENV=1      # A/B sweeps drive the launch-plan options through the environment (read only with this set)
# epoch-length sweep of one tools/bench_configs.py configuration: usage tools/sweep_rows.sh cfg4 1443 2886 5772
c=$1; shift
for r in "$@"; do
  GEOAC_S_ROWS=$r python tools/bench_configs.py $c 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rows $r:', round(d['seconds'],3), 's', '%.3e' % d['ray_steps_per_s'], 'launches', d.get('roofline',{}).get('launches_per_pass'))"
done
