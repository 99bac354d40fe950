"""geoac_amd - MI355X-native batched ray-fan integrator behind GeoAc's hot path.

The product is libgeoac_hip.so (hand-written gfx950 HIP kernels + a plain C ABI, include/geoac_hip.h).
This package is the thin Python mirror of that ABI used by the tests, bench.py and multi-GPU drivers;
it never computes rays itself and has no CPU fallback: without the built library, or without a GPU,
every compute call raises.
"""
from .api import (  # noqa: F401
    EQ_2D, EQ_3D, EQ_GLOBAL, EQ_3D_RNGDEP, EQ_GLOBAL_RNGDEP, EIG, EIG_STRIDE, REC, REC_STRIDE, FAN_STEP_LIMIT, FAN_SUB_FALLBACK, FAN_ABS_FALLBACK, GeoAcError, Params, FanContext, FanPool, load_library, library_path,
    met_load, natural_spline_slopes, fan_enumerate, default_params, options, option_names, has_ab_kernels, build_id, DEFAULT_OPTIONS,
)
