#!/usr/bin/env python3
"""bench.py - RK4 ray-steps/sec of the GeoAcGlobal ToyAtmo launch-angle fan on MI355X.

One "step" of this bench = one full pass of the hot path over the metric fan:
    GeoAcGlobal -prop ToyAtmo.met phi_min=-180 phi_max=179 phi_step=1  (theta 0.5..45/0.5, bounces=2,
    CalcAmp=True, WriteRays=False, lat_src=30 lon_src=0, rng_max=1500)   -> 360 az x 90 incl = 32 400 rays
(BASELINE.json metric / SURVEY.md §8d item 3).  Profile tables and launch angles are resident in HBM before
the timed region; the timed region covers RK4 + post-pass kernels, and for N > 1 the RCCL gather of arrivals.

N > 1 (one process per GPU, torch.distributed/RCCL): weak scaling by azimuth - the fan has N x 360 azimuths
(phi step 1/N degree), rank r integrates azimuth indices r, r+N, ...; arrival records are all-gathered over xGMI
and the step counts all-reduced.  value = all ranks' RK4 ray-steps / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
MET = os.path.join(ROOT, "tests", "golden", "ToyAtmo.met")

B_ALG_PER_STEP = 144          # 8 B x E(=18) state row per accepted RK4 step (Global, CalcAmp on)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s


def pmc_traffic_per_step():
    """HBM bytes per ray-step of the dominant kernel from the newest committed rocprofv3 PMC summary
    (profiles/rNN_*_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE passes of this same command); None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as fh:
            return float(json.load(fh)["k_rk4_hbm_bytes_per_ray_step"]), os.path.basename(files[-1])
    except Exception:
        return None, None


FP64_PEAK_TFLOPS = 78.6       # MI355X FP64 vector peak


def pmc_fp64_flop_per_step():
    """executed FP64 lane-flops per ray-step of the dominant kernel, from the SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 pass of the
    newest committed PMC summary (the binding resource of this path is FP64 VALU, not HBM); None if absent."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(f) as fh:
                v = json.load(fh).get("k_rk4_fp64_flop_per_ray_step")
            if v:
                return float(v), os.path.basename(f)
        except Exception:
            pass
    return None, None


def _cpu_worker(rank, n_workers, barrier, out):
    """one host core's share of the whole-socket baseline: its own process (the reference keeps its state in globals), two azimuth
    slices of the metric fan"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness as H
    import numpy as np
    phis = [-180.0 + 360.0 * (rank + n_workers * j) / (2 * n_workers) for j in range(2)]
    th = np.concatenate([H.fan_angles(phi_min=p, phi_max=p)[0] for p in phis])
    ph = np.concatenate([H.fan_angles(phi_min=p, phi_max=p)[1] for p in phis])
    cfg = H.make_cfg(H.EQ_GLOBAL, bounces=2, calc_amp=True, mode=0)
    lib = H.RefShim(H.EQ_GLOBAL, MET) if H.ref_available(H.EQ_GLOBAL) else H.Oracle(H.EQ_GLOBAL, MET)
    barrier.wait(timeout=300)
    t0 = time.perf_counter()
    steps, _, _, _ = lib.fan(cfg, th, ph)
    out.put((int(steps), time.perf_counter() - t0))


def cpu_baseline():
    """the reference's own serial loop on this box's host cores: three azimuth slices of the metric fan
    (270 rays, ~6e6 steps, 10-20 s), compiled reference if its prebuilt shim is present, else the plain-C port; then the same
    on every core this process may use, one process per core on disjoint azimuth slices (SURVEY §8d "whole-socket figure")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness as H
    th, ph = H.fan_angles(phi_min=-90.0, phi_max=90.0, phi_step=90.0)     # three azimuth slices (-90, 0, 90) of the metric fan
    cfg = H.make_cfg(H.EQ_GLOBAL, bounces=2, calc_amp=True, mode=0)
    if H.ref_available(H.EQ_GLOBAL):
        lib, kind = H.RefShim(H.EQ_GLOBAL, MET), "reference"
    else:
        lib, kind = H.Oracle(H.EQ_GLOBAL, MET), "port"
    t0 = time.perf_counter()
    steps, _, _, _ = lib.fan(cfg, th, ph)
    dt = time.perf_counter() - t0
    res = {"value": steps / dt, "unit": "RK4 ray-steps/s", "cores": 1, "kind": kind,
           "sample": f"phi = -90, 0, 90 slices of the metric fan: {len(th)} rays, {steps} steps, {dt:.1f} s "
                     f"({'compiled reference TUs -O2 (oracle/_ref)' if kind == 'reference' else 'plain-C restatement (oracle/)'})"}
    try:
        import multiprocessing as mp
        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:                                             # a container's CPU share (cgroup quota) is what is really there
            q = open("/sys/fs/cgroup/cpu.max").read().split()
            if q[0] != "max":
                n = min(n, max(1, int(float(q[0]) / float(q[1]) + 0.5)))
        except Exception:
            try:
                quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            except Exception:
                pass
        n = max(1, min(n, 64))
        ctx = mp.get_context("spawn")                    # fresh interpreters: nothing of this process's GPU state is inherited
        barrier, out = ctx.Barrier(n), ctx.Queue()
        procs = [ctx.Process(target=_cpu_worker, args=(r, n, barrier, out)) for r in range(n)]
        for p in procs:
            p.start()
        import queue
        got, deadline = [], time.time() + 900
        while len(got) < n and time.time() < deadline:
            try:
                got.append(out.get(timeout=1.0))
            except queue.Empty:
                if not any(p.is_alive() for p in procs) and out.empty():
                    break
        for p in procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        if len(got) < n:
            raise RuntimeError(f"{n - len(got)} of {n} baseline workers did not report")
        tot, worst = sum(g[0] for g in got), max(g[1] for g in got)
        res["all_cores"] = {"value": tot / worst, "unit": "RK4 ray-steps/s", "cores": n, "kind": kind,
                            "sample": f"{n} processes x 2 azimuth slices of the metric fan (180 rays each), {tot} steps, slowest {worst:.1f} s"}
    except Exception as e:                               # a baseline, not the product: report why it is missing
        res["all_cores"] = {"value": None, "error": repr(e)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phi-step", type=float, default=1.0, help="azimuth step of the N=1 fan (metric: 1.0)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the measured path) or gloo (rehearsal of the N>1 flow on fewer GPUs than ranks)")
    args = ap.parse_args()

    # CPU baseline first: its worker processes are started before this process has touched the GPU (and the host cores are idle
    # again before the timed region starts)
    cpu = None
    if not args.no_cpu_baseline and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        cpu = cpu_baseline()

    import numpy as np
    import torch
    import torch.distributed as dist
    import geoac_amd as G

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = min(local_rank, torch.cuda.device_count() - 1) if world > 1 else 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=args.backend)
    else:
        torch.cuda.set_device(0)
    n_gpus = world
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")

    # ---- the fan: 360*N azimuths x 90 inclinations, this rank's azimuth shard ----
    phi_step = args.phi_step / n_gpus
    th_all, ph_all = G.fan_enumerate(phi_min=-180.0, phi_max=180.0 - phi_step * 0.999999, phi_step=phi_step)
    n_theta = 90
    n_az = len(th_all) // n_theta
    from geoac_amd.sharding import gather_records, shard_by_azimuth
    theta, phi, _ = shard_by_azimuth(th_all, ph_all, n_theta, rank, n_gpus)

    stream = torch.cuda.current_stream(dev)
    ctx = G.FanContext(G.EQ_GLOBAL, device=dev.index, stream=stream.cuda_stream)
    ctx.load_met(MET)
    ctx.set_params(bounces=2, calc_amp=1, mode=0)
    ctx.set_angles(theta, phi)                                 # inputs resident in HBM before timing
    legs = 3
    rec_local = torch.empty((len(theta), legs, G.REC_STRIDE), dtype=torch.float64, device=dev)
    steps_t = torch.zeros(1, dtype=torch.int64, device=coll_dev)
    gathered = {}

    def one_pass():
        ctx.launch()
        if world > 1:
            ctx.copy_records_to(rec_local.data_ptr())          # D2D on the context's (= torch's current) stream
            # gather of the arrival records of the whole fan: RCCL all_gather over xGMI (gloo: host rehearsal)
            gathered["rec"] = gather_records(rec_local if coll_dev == dev else rec_local.cpu(), n_az, n_theta)
            steps_t[0] = ctx.total_steps()
            dist.all_reduce(steps_t)
            return int(steps_t.item())
        return ctx.total_steps()

    for _ in range(args.warmup):
        one_pass()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total_steps = 0
    rk4_ms = post_ms = 0.0
    rk4_launches = 0
    for _ in range(args.steps):
        total_steps += one_pass()
        tm = ctx.timing()
        rk4_ms += tm["ms_rk4"]; post_ms += tm["ms_post"]; rk4_launches += tm["epochs"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every leg that produced a row somewhere must be present in the gathered table
        assert gathered["rec"].shape[0] == n_az * n_theta

    local_steps_per_pass = ctx.total_steps()
    if rank == 0:
        value = total_steps / dt
        # dominant kernel = k_rk4: algorithmic bytes per launch / average launch duration (HIP events on the kernel's own stream,
        # recorded inside libgeoac_hip around every k_rk4 epoch of the timed passes).  One epoch of this fan is TWO concurrent k_rk4
        # launches of the same duration by construction (the shallow tenth of the rays on k_rk4<EqGlobalPair>, the rest on
        # k_rk4<EqGlobal<true>> with 0.75 x the rows); "launch" below = that pair, bytes and steps are those of both.
        ach_gbs = (B_ALG_PER_STEP * local_steps_per_pass * args.steps) / (rk4_ms * 1e-3) / 1e9 if rk4_ms > 0 else 0.0
        bps, pmc_src = pmc_traffic_per_step()
        steps_per_launch = local_steps_per_pass * args.steps / max(rk4_launches, 1)
        out = {
            "metric": "RK4 ray-steps/sec, GeoAcGlobal 360x90 ToyAtmo fan; arrivals within 1e-6 of ref",
            "value": value, "unit": "RK4 ray-steps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"GeoAcGlobal -prop ToyAtmo.met, {n_az} az x {n_theta} incl = {n_az * n_theta} rays "
                                   f"(phi step {phi_step:g} deg), bounces=2, CalcAmp=True, WriteRays=False, rng_max=1500",
                       "rays_per_gpu": int(len(theta)), "ray_steps_per_pass": int(total_steps // args.steps),
                       "parallelism": f"azimuth-sharded x{n_gpus}" + (f", {'RCCL' if args.backend == 'nccl' else args.backend} all_gather of arrivals" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS,
                         "traffic": (bps * steps_per_launch) if bps else None, "traffic_unit": "bytes per launch",
                         "traffic_source": pmc_src, "achieved_bytes_per_launch": B_ALG_PER_STEP * steps_per_launch,
                         "kernel": "k_rk4<EqGlobalPair,true,false> || k_rk4<EqGlobal<true>,true,false> (one epoch)", "launches": rk4_launches,
                         "avg_launch_ms": rk4_ms / max(rk4_launches, 1),
                         "alg_bytes_per_step": B_ALG_PER_STEP,
                         "rk4_ms_per_pass": rk4_ms / args.steps, "postpass_ms_per_pass": post_ms / args.steps},
        }
        fps, fp_src = pmc_fp64_flop_per_step()
        if fps and rk4_ms > 0:
            tf = fps * local_steps_per_pass * args.steps / (rk4_ms * 1e-3) / 1e12
            # the honest "how busy is the binding unit" figure beside the contract's HBM roofline
            out["roofline"]["fp64"] = {"flop_per_step": fps, "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": tf / FP64_PEAK_TFLOPS, "source": fp_src}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
