"""A/B of two builds of libgeoac_hip.so on configurations of tools/bench_configs.py: usage ab_cfg.py <libA.so> <libB.so> cfg [cfg ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = "import sys, os; sys.path.insert(0, %r); import geoac_amd.api as A; A.library_path = lambda: %r; sys.argv = ['x'] + %r; __file__ = os.path.join(%r, 'tools', 'bench_configs.py'); exec(open(__file__).read())"
for lib in (sys.argv[1], sys.argv[2], sys.argv[1], sys.argv[2]):
    r = subprocess.run([sys.executable, "-c", code % (ROOT, os.path.abspath(lib), sys.argv[3:], ROOT)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    for l in r.stdout.decode().strip().split("\n"):
        try:
            d = json.loads(l); print(os.path.basename(lib), d["config"][:28], "%.4f s  %.4e steps/s" % (d["seconds"], d["ray_steps_per_s"]), flush=True)
        except Exception:
            print(os.path.basename(lib), l[:200])
