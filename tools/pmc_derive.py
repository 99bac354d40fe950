"""Per-ray-step figures of the RK4 and post-pass kernels from a tools/pmc_summary.py JSON.
usage: pmc_derive.py <pmc_summary.json> <ray_steps_in_the_profiled_run> <out_pmc_traffic.json>
All k_rk4<...> instantiations of the run are summed (a hybrid fan runs k_rk4<EqGlobalPair> and k_rk4<EqGlobal> side by side)."""
import json, sys

summ, steps, out = json.load(open(sys.argv[1])), float(sys.argv[2]), sys.argv[3]


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
    "k_rk4_hbm_bytes_per_ray_step": (tot("k_rk4", "FETCH_SIZE") + tot("k_rk4", "WRITE_SIZE")) / steps,
    "k_postpass_hbm_bytes_per_ray_step": (tot("k_postpass", "FETCH_SIZE") + tot("k_postpass", "WRITE_SIZE")) / steps,
    "k_accum_hbm_bytes_per_ray_step": (tot("k_accum", "FETCH_SIZE") + tot("k_accum", "WRITE_SIZE")) / steps,
    "fp64_vector_peak_tflops": 78.6,
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
