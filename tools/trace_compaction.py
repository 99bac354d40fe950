"""Per-epoch live rays / live waves of a fan with and without live-ray compaction (GEOAC_TRACE_EPOCHS output of libgeoac_hip), config 3
by default: how full the launched waves are in the late epochs.  usage: trace_compaction.py [cfg3|cfg4]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
for compact in ("1", "0"):
    env = dict(os.environ, GEOAC_DEBUG_ENV="1", GEOAC_TRACE_EPOCHS="1", GEOAC_COMPACT=compact)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_configs.py"), which], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    lines = [l for l in r.stderr.decode().split("\n") if l.startswith("[epoch")]
    n = len(lines) // 2                       # bench_configs runs a warm-up and a timed pass: keep the second
    print(f"== {which}, GEOAC_COMPACT={compact}: {r.stdout.decode().strip()[:0]}")
    for l in lines[n:]:
        print("  ", l)
    import json
    d = json.loads(r.stdout.decode().strip().split("\n")[-1])
    print(f"   -> {d['seconds']:.3f} s, {d['ray_steps_per_s']:.3e} ray-steps/s")
