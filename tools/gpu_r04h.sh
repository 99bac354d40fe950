cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
export GEOAC_AB_SET=cfg3 GEOAC_DEBUG_ENV=1
for rep in 1 2; do
GEOAC_AB_TAG="r03" python tools/ab_metric.py 3 build_ab_r03/libgeoac_hip.so 2>&1 | tail -1
GEOAC_RK4_PREFETCH=1 GEOAC_AB_TAG="prefetch+skip" python tools/ab_metric.py 3 geoac_amd/libgeoac_hip.so 2>&1 | tail -1
GEOAC_RK4_PREFETCH=0 GEOAC_AB_TAG="noprefetch+uncond" python tools/ab_metric.py 3 geoac_amd/libgeoac_hip.so 2>&1 | tail -1
GEOAC_RK4_PREFETCH=1 GEOAC_AB_TAG="prefetch+uncond" python tools/ab_metric.py 3 build_ab_pfu/libgeoac_hip.so 2>&1 | tail -1
done > gpurun_out/r04h/cfg3_4way.log 2>&1
cat gpurun_out/r04h/cfg3_4way.log
