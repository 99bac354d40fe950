"""Per-ray-step figures of the RK4 and post-pass kernels from a tools/pmc_summary.py JSON.
usage: pmc_derive.py <pmc_summary.json> <ray_steps_in_the_profiled_run> <out_pmc_traffic.json> [<libgeoac_hip.so that was profiled>]
All k_rk4<...> instantiations of the run are summed (a hybrid fan runs k_rk4<EqGlobalPair> and k_rk4<EqGlobal> side by side).

HBM bytes: FETCH_SIZE and WRITE_SIZE come from separate passes.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read
(MI355X_MICROARCH.md, HBM: the counter tallies 128-B requests at 64 B), WRITE_SIZE is exact: read bytes = 2 x FETCH_SIZE.  The guide calibrated the factor on
16-B-per-lane accesses; these kernels read 8 B per lane (512 contiguous bytes per wave instruction: whole 128-B lines), so the corrected figure is the one to
compare with a byte count and the raw one is kept beside it.  `lib_build_id`: the build the counters belong to (geoac_build_id: a hash of its sources and flags; `lib_sha256` is the file's,
which a rebuild of the same tree changes) - bench.py marks the figures stale when the library it loaded is another build."""
import hashlib, json, sys

summ, steps, out = json.load(open(sys.argv[1])), float(sys.argv[2]), sys.argv[3]
lib = sys.argv[4] if len(sys.argv) > 4 else None


def build_id(path):
    """geoac_build_id() of the profiled library: the hash of the sources and flags it was compiled from (a rebuild of the same tree is the same build)"""
    import ctypes
    L = ctypes.CDLL(path)
    L.geoac_build_id.restype = ctypes.c_char_p
    return L.geoac_build_id().decode()
FETCH_CORRECTION = 2.0


def tot(prefix, counter):
    return sum(v[counter]["total"] for k, v in summ.items() if k.startswith(prefix) and counter in v)


def flops(prefix):
    return 64.0 * (tot(prefix, "SQ_INSTS_VALU_ADD_F64") + tot(prefix, "SQ_INSTS_VALU_MUL_F64")
                   + 2.0 * tot(prefix, "SQ_INSTS_VALU_FMA_F64") + tot(prefix, "SQ_INSTS_VALU_TRANS_F64"))


res = {
    "command": "rocprofv3 --kernel-trace --pmc <set> -d gpurun_out/pmc/<set> --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
               "--no-cpu-baseline; four separate passes: {ADD,MUL,FMA,TRANS}_F64 | SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS | "
               "FETCH_SIZE | WRITE_SIZE; aggregated by tools/pmc_summary.py, reduced by tools/pmc_derive.py",
    "ray_steps_in_pass": steps,
    "note": "SQ_INSTS_VALU_*_F64 count wave instructions; executed lane-flops = 64 x (ADD + MUL + 2 FMA + TRANS), exec-masked and redundant "
            "lanes included (the two-lane kernel integrates the base ray in both lanes of a pair); k_rk4 = all k_rk4 instantiations of the run",
    "k_rk4_kernels": sorted(k for k in summ if k.startswith("k_rk4")),
    "k_rk4_fp64_flop_per_ray_step": flops("k_rk4") / steps,
    "k_postpass_fp64_flop_per_ray_step": flops("k_postpass") / steps,
    "k_rk4_valu_insts_per_ray_step": tot("k_rk4", "SQ_INSTS_VALU") / steps,
    "fetch_size_correction": FETCH_CORRECTION,
    "k_rk4_hbm_bytes_per_ray_step": (FETCH_CORRECTION * tot("k_rk4", "FETCH_SIZE") + tot("k_rk4", "WRITE_SIZE")) / steps,
    "k_postpass_hbm_bytes_per_ray_step": (FETCH_CORRECTION * tot("k_postpass", "FETCH_SIZE") + tot("k_postpass", "WRITE_SIZE")) / steps,
    "k_accum_hbm_bytes_per_ray_step": (FETCH_CORRECTION * tot("k_accum", "FETCH_SIZE") + tot("k_accum", "WRITE_SIZE")) / steps,
    "all_kernels_hbm_bytes_per_ray_step": (FETCH_CORRECTION * tot("", "FETCH_SIZE") + tot("", "WRITE_SIZE")) / steps,
    "raw_counters_per_ray_step": {k: {"FETCH_SIZE": tot(k, "FETCH_SIZE") / steps, "WRITE_SIZE": tot(k, "WRITE_SIZE") / steps} for k in ("k_rk4", "k_postpass", "k_accum")},
    "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest() if lib else None,
    "lib_build_id": build_id(lib) if lib else None,
    "fp64_vector_peak_tflops": 78.6,
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
