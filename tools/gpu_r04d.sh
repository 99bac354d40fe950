cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
export GEOAC_AB_SET=cfg3
python tools/ab_metric.py 3 build_ab_pf0/libgeoac_hip.so geoac_amd/libgeoac_hip.so build_ab_pf2/libgeoac_hip.so > gpurun_out/r04d/cfg3_pf.log 2>&1; cat gpurun_out/r04d/cfg3_pf.log
unset GEOAC_AB_SET
python tools/ab_metric.py 6 build_ab_pf0/libgeoac_hip.so geoac_amd/libgeoac_hip.so build_ab_pf2/libgeoac_hip.so > gpurun_out/r04d/metric_pf.log 2>&1; cat gpurun_out/r04d/metric_pf.log
GEOAC_AB_SET=3d python tools/ab_metric.py 4 build_ab_pf0/libgeoac_hip.so geoac_amd/libgeoac_hip.so > gpurun_out/r04d/cfg2_pf.log 2>&1; cat gpurun_out/r04d/cfg2_pf.log
python - > gpurun_out/r04d/cfg4_check.log 2>&1 <<'PY'
import sys, os, traceback
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, tempfile
import geoac_amd as G
import rngdep_data as RD
import bench
grid = RD.write_grid(os.path.join(tempfile.gettempdir(), "g4"), short_paths=False, thin=1)
th4, ph4 = G.fan_enumerate(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 999 * 0.36, phi_step=0.36)
dev = torch.device("cuda", 0)
r4 = bench.FanRun(G, G.EQ_3D_RNGDEP, lambda c: c.load_grid(*grid), dict(bounces=1, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0)), th4, ph4, int(np.sum(ph4 == ph4[0])), 0, 1, dev, dev, torch.cuda.current_stream(dev).cuda_stream, 2, False)
s = r4.one_pass()
try:
    print(bench.check_config4(r4, s))
except Exception:
    traceback.print_exc()
PY
tail -30 gpurun_out/r04d/cfg4_check.log
