#!/usr/bin/env python3
"""bench.py - RK4 ray-steps/sec of the GeoAcGlobal ToyAtmo launch-angle fan on MI355X.

One "step" of this bench = one full pass of the hot path over the metric fan:
    GeoAcGlobal -prop ToyAtmo.met phi_min=-180 phi_max=179 phi_step=1  (theta 0.5..45/0.5, bounces=2,
    CalcAmp=True, WriteRays=False, lat_src=30 lon_src=0, rng_max=1500)   -> 360 az x 90 incl = 32 400 rays
(BASELINE.json metric / SURVEY.md §8d item 3).  The profile tables are resident in HBM before the timed region (atmosphere upload
excluded); a timed pass is what geoac_fan_run does: launch angles host -> device, every kernel of the fan (RK4 epochs,
post-pass, sums), arrival records device -> host - and for N > 1 the RCCL gather of the arrival tables in between.
`launch_only` repeats the figure without the two copies (HIP events around the kernels).

Parity gate before timing: the warm-up pass's records are compared with tests/golden/full_metric.npz, the fan as the COMPILED
REFERENCE integrated it (tests/golden/make_golden_full.py): step count and outcome of every (ray, leg) exact, travel time /
attenuation / turning height / arrival angles / amplitude / range within 1e-6.  A failed gate aborts the bench.

N > 1 (one process per GPU, torch.distributed/RCCL).  `--scaling weak` (default; the line's `value`): the fan has N x 360 azimuths
(phi step 1/N degree), rank r integrates azimuth indices r, r+N, ... - per-GPU work fixed.  `--scaling strong`: the metric fan itself,
360 azimuths dealt round-robin to the ranks.  Whatever the line is, the other flavour and the strong-scaling run of BASELINE's
config 4 (GeoAc3D.RngDep, 5x5x1400 grid, 1000 az x 1000 incl = 1 M rays sharded by azimuth) are measured in the same run and
reported under "other_scalings" (skip with --no-extras).  Arrival records are all-gathered over xGMI and the step counts all-reduced.
value = all ranks' RK4 ray-steps / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
MET = os.path.join(ROOT, "tests", "golden", "ToyAtmo.met")

B_ALG_PER_STEP = 144          # 8 B x E(=18) state row per accepted RK4 step (Global, CalcAmp on)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6       # MI355X FP64 vector peak


def newest_pmc():
    """per-ray-step figures of the dominant kernel from the newest committed rocprofv3 PMC summary (profiles/rNN_*_pmc_traffic.json:
    separate FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU_*_F64 passes of this same command, tools/profile_round.sh; FETCH_SIZE x 2 as
    MI355X_MICROARCH.md prescribes for gfx950); {} if absent.  Counters cannot be collected inside this run, so the file names the code
    build it was measured on (lib_build_id = geoac_build_id(): a hash of the library's sources and flags; hipcc's output is not
    bit-reproducible, so a file hash would call a rebuild of the same tree another build): `_stale` is True when the library loaded now is another build."""
    import glob
    import hashlib
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(f) as fh:
                d = json.load(fh)
            d["_source"] = os.path.basename(f)
            import geoac_amd
            now = geoac_amd.build_id()                          # hash of the sources and flags the loaded library was compiled from
            d["_stale"] = (d.get("lib_build_id") != now)
            d["_build_id_now"] = now
            return d
        except Exception:
            pass
    return {}


def newest_isa_mix():
    """static instruction count of the two-lane RK4 kernel's step loop (tools/isa_mix.py --trace on hipcc's assembly of this source,
    committed as profiles/rNN_*_isa_mix_pair.json); {} if absent"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_mix_pair.json")), reverse=True):
        try:
            with open(f) as fh:
                d = json.load(fh)
            d["_source"] = os.path.basename(f)
            return d
        except Exception:
            pass
    return {}


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


class FanRun:
    """one azimuth-sharded fan on this rank's GPU: pass() = angles H2D + all kernels + (N > 1: RCCL gather) + records D2H"""

    def __init__(self, G, eq, load, params, th_all, ph_all, n_theta, rank, world, dev, coll_dev, stream, legs, collective=None):
        import numpy as np
        import torch
        from geoac_amd.sharding import shard_by_azimuth
        self.G, self.world, self.rank, self.dev, self.coll_dev, self.legs = G, world, rank, dev, coll_dev, legs
        self.collective = (world > 1) if collective is None else collective       # gather / reduce through the process group (N > 1; --force-collective)
        self.n_theta, self.n_az = n_theta, len(th_all) // n_theta
        self.theta_all, self.phi_all = th_all, ph_all
        self.theta, self.phi, self.idx = shard_by_azimuth(th_all, ph_all, n_theta, rank, world)
        self.ctx = G.FanContext(eq, device=dev.index, stream=stream)
        load(self.ctx)
        self.ctx.set_params(**params)
        self.rec_local = torch.empty((len(self.theta), legs, G.REC_STRIDE), dtype=torch.float64, device=dev) if self.collective else None
        # the arrivals' landing place on the host: pinned, so the device -> host copy runs at link speed
        self.rec_host = torch.empty((len(self.theta), legs, G.REC_STRIDE), dtype=torch.float64, pin_memory=True)
        self.steps_t = torch.zeros(1, dtype=torch.int64, device=coll_dev)
        self.rec = None            # N = 1: this pass's records; N > 1, rank 0, when want_full: the whole fan's gathered table (parity gate)
        self.want_full = False
        self.np, self.torch = np, torch

    def one_pass(self):
        import torch.distributed as dist
        from geoac_amd.sharding import gather_records
        ctx = self.ctx
        if not self.collective:
            self.rec, steps = ctx.run(self.theta, self.phi, out=self.rec_host.numpy())    # set_angles + launch + fetch (geoac_fan_run)
            return steps
        # a rank that cannot run its share (out of memory on a shared device, ...) must not leave the others waiting in the gather: every rank
        # learns about it before any data collective and all of them raise
        err = None
        try:
            ctx.set_angles(self.theta, self.phi)
            ctx.launch()
        except Exception as e:                                           # noqa: BLE001
            err = e
        ok = self.torch.tensor([0 if err else 1], dtype=self.torch.int64, device=self.coll_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            raise RuntimeError(f"rank {self.rank}: {err!r}" if err else "another rank failed to launch its share of the fan")
        ctx.copy_records_to(self.rec_local.data_ptr())                  # D2D on the context's (= torch's current) stream
        # the whole fan's table on every GPU: RCCL all_gather over xGMI (gloo: host rehearsal)
        full = gather_records(self.rec_local if self.coll_dev == self.dev else self.rec_local.cpu(), self.n_az, self.n_theta)
        self.steps_t[0] = ctx.total_steps()
        dist.all_reduce(self.steps_t)
        # arrivals device -> host: every rank lands its own share (the N processes' pinned buffers together hold the fan, as the one
        # buffer does at N = 1); the gathered table goes to the host only for the parity gate
        self.rec_host.copy_(self.rec_local, non_blocking=True)
        self.torch.cuda.current_stream().synchronize()
        if self.rank == 0 and self.want_full:
            self.rec = full.cpu().numpy()
        return int(self.steps_t.item())

    def timed(self, n_pass):
        """barrier + sync, n_pass passes, sync + barrier; returns (ray-steps, max-over-ranks seconds, kernel-event sums)"""
        import torch.distributed as dist
        torch = self.torch
        if self.collective:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total = 0
        ev = {"ms_total": 0.0, "ms_rk4": 0.0, "ms_post": 0.0, "epochs": 0}
        for _ in range(n_pass):
            total += self.one_pass()
            tm = self.ctx.timing()
            for k in ev:
                ev[k] += tm[k]
        torch.cuda.synchronize()
        if self.collective:
            dist.barrier()
        dt = time.perf_counter() - t0
        if self.collective:
            t = torch.tensor([dt], dtype=torch.float64, device=self.coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return total, dt, ev


def agree(ok, collective, coll_dev, what, err=None):
    """every rank learns whether every rank got through `what`; all of them raise together if one did not - a rank that failed alone would
    otherwise skip the collectives the others are about to enter (a hang until the RCCL timeout, or mismatched operations)"""
    if collective:
        import torch
        import torch.distributed as dist
        t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        all_ok = int(t.item()) == 1
    else:
        all_ok = bool(ok)
    if not all_ok:
        raise RuntimeError(f"{what}: {err!r}" if err is not None else f"{what}: failed on another rank")


def parity_gate(rec, phi_step_mult):
    """rank 0: the fan's records against the reference-made fixture of the metric fan; with N x 360 azimuths every N-th one is a
    fixture azimuth.  Returns a dict for the JSON line; raises on a mismatch."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "full_metric.npz")
    if not os.path.exists(path):
        return {"status": "skipped", "why": "tests/golden/full_metric.npz not present"}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity import compare_compact
    g = np.load(path)
    n_theta = 90
    az = np.arange(360) * phi_step_mult
    idx = (az[:, None] * n_theta + np.arange(n_theta)[None, :]).reshape(-1)
    err = compare_compact(rec, g, idx=idx)
    return {"status": "pass", "fixture": "tests/golden/full_metric.npz (compiled reference, every ray of the 360 x 90 fan)",
            "rays_checked": int(len(idx)), "legs_checked": int(len(idx) * 3), "counts": "exact",
            "max_rel_err": {k: v for k, v in err.items() if k != "AMP_exempt"},
            "amp_rule": "AMP within 1e-6 except the arrivals the fixture names (amp_exempt: ill-conditioned in the compiled reference itself, "
                        "at most max(3, 1e-4 N)), those within 4 x the reference's own sensitivity; an unnamed loose arrival fails the gate",
            "amp_exempt": [{"ray": int(idx[r]), "theta_index": int(idx[r] % n_theta), "azimuth_index": int(idx[r] // n_theta), "leg": l, "rel_err": e, "bound": b}
                           for r, l, e, b in err.get("AMP_exempt", [])]}


def check_config4(r4, steps):
    """rank 0: the WHOLE 999 x 1000 config-4 fan's records (N > 1: the gathered table) checked - size-independent properties of every ray
    (count bookkeeping, monotone sums, eikonal residual |nu| c(arrival) / c0 = 1 at every arrival, c through the device-function probe of
    include/geoac_probe.h, which tests/test_gpu_probes.py pins to the reference's interpolant), and every ray the compiled reference
    integrated (tests/golden/full_cfg4*.npz: the 2 000-ray lattice of the rank-0 share, 250 rays of each other rank's share, the
    10 000-ray lattice of the whole fan): step counts and flags exact, values within 1e-6"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity import compare_compact, fan_properties
    G = r4.G
    rec = r4.rec
    REC = G.REC
    valid = rec[..., REC["VALID"]] > 0
    st = rec[..., REC["STATE"]:REC["STATE"] + 3][valid]
    _, a7 = r4.ctx.probe_grid(st[:, 0], st[:, 1], st[:, 2])
    _, a0 = r4.ctx.probe_grid(np.zeros(1), np.zeros(1), np.zeros(1))
    narr = fan_properties(rec, steps, 18, slice(3, 6), c_ratio=a0[0, 0] / a7[:, 0], eik_tol=5e-3)
    out = {"rays": int(rec.shape[0]), "arrivals_eikonal_checked": narr, "ray_steps": int(steps), "vs_reference": {}}
    gd = os.path.join(ROOT, "tests", "golden")
    worst, n_ref = {}, 0
    for f in ("full_cfg4.npz", "full_cfg4_lattice.npz"):
        if os.path.exists(os.path.join(gd, f)):
            g = np.load(os.path.join(gd, f))
            assert np.array_equal(r4.theta_all[g["sel"]], g["theta"]) and np.array_equal(r4.phi_all[g["sel"]], g["phi"]), f"{f}: the fan's launch angles are not the fixture's"
            err = compare_compact(rec, {k: g[k] for k in ("steps", "flags", "vals", "val_fields")}, idx=g["sel"])
            n_ref += len(g["sel"])
            for k, v in err.items():
                if not isinstance(v, list):
                    worst[k] = max(worst.get(k, 0.0), v)
    f = os.path.join(gd, "full_cfg4_shares.npz")
    if os.path.exists(f):
        g = np.load(f)
        for r in range(1, 8):
            err = compare_compact(rec, {"steps": g[f"steps{r}"], "flags": g[f"flags{r}"], "vals": g[f"vals{r}"], "val_fields": g["val_fields"]}, idx=g[f"sel{r}"])
            n_ref += len(g[f"sel{r}"])
            for k, v in err.items():
                if not isinstance(v, list):
                    worst[k] = max(worst.get(k, 0.0), v)
    out["vs_reference"] = {"rays": n_ref, "counts": "exact", "max_rel_err": worst, "fixtures": "tests/golden/full_cfg4{,_lattice,_shares}.npz (compiled reference)"}
    return out


def config5_ring(G, args, rank, world, dev, coll_dev, collective, build, agree):
    """BASELINE config 5 as a strong-scaling run: GeoAcGlobal.RngDep -eig_search (bounces 0..2) to the 64 receivers of the 2.5-degree
    ring, receivers dealt round robin to the ranks (geoac_amd.sharding.shard_receivers), every rank one batched geoac_eig_search on its
    share, the eigenray tables all-gathered (gather_eigenrays; RCCL when the tables live on GPUs); rank 0 checks the gathered table against
    the reference binary's result files (tests/golden/cli/cfg5_*: every ring position that has one) and its own receivers' iteration
    logs line for line.  One untimed search (contexts, clones, buffers), one timed."""
    import tempfile
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rngdep_data as RD
    from geoac_amd.sharding import gather_eigenrays, shard_receivers
    from parity import compare_eig_ring, compare_logs, ring_golden_name, ring_receivers
    n_rcvr = 64
    rc = ring_receivers(n_rcvr)
    mine = shard_receivers(n_rcvr, rank, world)

    def make():
        ctx = G.FanContext(G.EQ_GLOBAL_RNGDEP, device=dev.index)
        ctx.load_grid(*RD.write_grid_global(tempfile.mkdtemp(prefix=f"bench_ring_{rank}_"), short_paths=False))
        ctx.set_params(src=(0.0, 31.0, 0.0))
        return ctx
    ctx = build("config-5 set-up", make)

    def search():
        res, err = None, None
        try:
            res = ctx.eig_search(rc[mine], bnc_min=0, bnc_max=2, verbose=True)
        except Exception as e:                                             # noqa: BLE001
            err = e
        agree(err is None, collective, coll_dev, "config-5 search", err)
        full = gather_eigenrays(torch.from_numpy(res["eig"]).to(coll_dev), mine)
        st = torch.tensor([res["stats"]["steps"], res["stats"]["rays"], res["stats"]["launches"]], dtype=torch.int64, device=coll_dev)
        if collective:
            dist.all_reduce(st)
        return res, full, st

    res, full, st = search()                                               # untimed
    gerr, chk = None, None
    if rank == 0:
        try:
            cli = os.path.join(ROOT, "tests", "golden", "cli")
            chk = compare_eig_ring(full.cpu().numpy(), range(n_rcvr), cli)
            logs = 0
            for k, p in enumerate(mine):
                f = os.path.join(cli, ring_golden_name(int(p)), "LOG.txt")
                if os.path.exists(f):
                    compare_logs(res["logs"][k], open(f).read())
                    logs += 1
            chk["rank0_logs_line_for_line"] = logs
            chk["eigenrays_gathered"] = int(full.shape[0])
        except Exception as e:                                             # noqa: BLE001
            gerr = e
    agree(gerr is None, collective, coll_dev, "config-5 check against the reference binary's files", gerr)
    if collective:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, full, st = search()
    torch.cuda.synchronize()
    if collective:
        dist.barrier()
    dt = time.perf_counter() - t0
    if collective:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    steps, rays, launches = (int(x) for x in st.tolist())
    ctx.close()
    return {"value": steps / dt, "unit": "RK4 ray-steps/s", "seconds": dt, "receivers": n_rcvr, "receivers_per_gpu": int(len(mine)), "rays": rays, "ray_steps": steps,
            "fan_launches_all_ranks": launches, "rounds_rank0": res["stats"]["rounds"], "scaling": "strong",
            "workload": "GeoAcGlobal.RngDep -eig_search, 5x5 grid, 64 receivers on a 2.5 deg ring, bounces 0..2, sharded by receiver; eigenray tables all-gathered",
            "note": "a search lasts rounds x the longest ray of each round: sharding receivers shortens the scan rounds' fans, not the chain of decision rounds",
            "checked": chk}


def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher, N > 1: this process starts the N ranks itself - the contract's launch line
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`) as a CHILD
    process, before anything here has touched a GPU (the CPU baseline, which runs first, is host-only; nothing that has initialised the
    GPU is ever replaced by another program).  The child's JSON line is relayed with this process's cpu_baseline merged in."""
    import socket
    import subprocess
    cpu = None if args.no_cpu_baseline else cpu_baseline()
    if args.backend == "nccl":
        import torch
        have = torch.cuda.device_count()                   # counts devices without initialising one
        if have < args.gpus:
            sys.exit(f"bench.py --gpus {args.gpus}: this node shows {have} GPU(s); RCCL needs one device per rank "
                     f"(a rehearsal of the N-rank flow on fewer devices: --backend gloo)")
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    argv = [a for a in sys.argv[1:] if a != "--no-cpu-baseline"] + ["--no-cpu-baseline"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    child = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in child.stdout:
        if out.startswith("{") and out.rstrip().endswith("}"):
            line = out                                      # the ranks' ONE JSON line (rank 0 prints it)
        else:
            sys.stderr.write(out)
    rc = child.wait()
    if rc != 0 or line is None:
        sys.exit(rc or 1)
    d = json.loads(line)
    d["launcher"] = f"bench.py --gpus {args.gpus} started its {args.gpus} ranks itself (child: torch.distributed.run, port {port})"
    if cpu is not None:
        d["cpu_baseline"] = cpu
    print(json.dumps(d), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak", help="which N > 1 fan the line's value is measured on")
    ap.add_argument("--no-extras", action="store_true", help="skip the other-scaling and config-4 measurements")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the measured path) or gloo (rehearsal of the N>1 flow on fewer GPUs than ranks)")
    ap.add_argument("--force-collective", action="store_true", help="take the N > 1 path (process group, RCCL all_gather / all_reduce of device tensors) at N = 1 too: "
                    "under torch.distributed.run --nproc-per-node 1 this executes the RCCL path on a one-GPU box (tests/test_gpu_nccl.py)")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ        # under torch.distributed.run (the driver's N > 1 line, or spawn_ranks' child)
    if args.gpus > 1 and not launched:
        return spawn_ranks(args)
    if launched and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {os.environ['WORLD_SIZE']} rank(s)")

    # CPU baseline first: its worker processes are started before this process has touched the GPU (and the host cores are idle
    # again before the timed region starts).  Under a launcher rank 0 measures it while the other ranks wait for it in the rendezvous.
    cpu = None
    if not args.no_cpu_baseline and int(os.environ.get("RANK", "0")) == 0:
        cpu = cpu_baseline()

    import numpy as np
    import torch
    import torch.distributed as dist
    import geoac_amd as G

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = min(local_rank, torch.cuda.device_count() - 1) if world > 1 else 0
    collective = world > 1 or args.force_collective
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
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
    stream = torch.cuda.current_stream(dev).cuda_stream

    def build(what, make):
        """rank-local preparation (angle lists, input files, a FanRun with its context and buffers) on every rank, or a failure on every
        rank: nothing in `make` is a collective, and every rank learns the others' outcome before any of them enters one"""
        r, err = None, None
        try:
            r = make()
        except Exception as e:                                             # noqa: BLE001
            err = e
        agree(err is None, collective, coll_dev, what, err)
        return r

    def metric_run(scaling):
        def make():
            # the whole fan for N ranks and the multiple of the fixture's azimuth step it is enumerated with
            mult = n_gpus if scaling == "weak" else 1
            step = 1.0 / mult
            th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=180.0 - step * 0.999999, phi_step=step)
            return FanRun(G, G.EQ_GLOBAL, lambda c: c.load_met(MET), dict(bounces=2, calc_amp=1, mode=0), th, ph, 90, rank, world, dev, coll_dev, stream, 3, collective), mult
        return build("metric fan set-up", make)

    # ---- the line's fan ----
    run, mult = metric_run(args.scaling)
    gate = {"status": "skipped", "why": "--warmup 0 (profiler passes): the gate runs on the first untimed pass"}
    for w in range(args.warmup):
        run.want_full = (w == 0)
        run.one_pass()
        if w == 0:
            gerr = None
            if rank == 0:
                try:
                    gate = parity_gate(run.rec, mult)                  # a mismatch: no timing without parity
                    run.longest_ray_steps = int(run.rec[..., G.REC["STEPS"]].sum(axis=1).max())
                except Exception as e:                                 # noqa: BLE001
                    gerr = e
            agree(gerr is None, collective, coll_dev, "parity gate", gerr)  # (every rank stops, not just rank 0)
    run.want_full = False
    total_steps, dt, ev = run.timed(args.steps)
    local_steps_per_pass = run.ctx.total_steps()

    extras = {}
    if not args.no_extras:
        try:
            other = "strong" if args.scaling == "weak" else "weak"
            if world > 1:
                r2, m2 = metric_run(other)
                r2.want_full = True
                r2.one_pass()
                r2.want_full = False
                gerr = None
                if rank == 0:
                    try:
                        parity_gate(r2.rec, m2)
                    except Exception as e:                              # noqa: BLE001
                        gerr = e
                agree(gerr is None, collective, coll_dev, f"parity gate of the {other}-scaling fan", gerr)
                s2, t2, _ = r2.timed(max(1, min(args.steps, 3)))
                extras[f"metric_fan_{other}"] = {"value": s2 / t2, "unit": "RK4 ray-steps/s", "ms_per_pass": t2 / max(1, min(args.steps, 3)) * 1e3,
                                                 "rays": int(r2.n_az * r2.n_theta), "rays_per_gpu": int(len(r2.theta)), "scaling": other,
                                                 "note": "a fan lasts as long as its longest ray (54 130 steps x the step latency): sharding the fixed 32 400-ray fan does not shorten it" if other == "strong" else "N x 360 azimuths"}
                del r2
            # BASELINE config 4: GeoAc3D.RngDep, 5x5x1400 grid, 1000 az x 1000 incl, strong scaling by azimuth
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import tempfile
            import rngdep_data as RD

            def make4():
                grid = RD.write_grid(os.path.join(tempfile.gettempdir(), f"bench_grid_{rank}"), short_paths=False, thin=1)
                th4, ph4 = G.fan_enumerate(theta_min=0.05, theta_max=50.0, theta_step=0.05, phi_min=-180.0, phi_max=-180.0 + 999 * 0.36, phi_step=0.36)
                return FanRun(G, G.EQ_3D_RNGDEP, lambda c: c.load_grid(*grid), dict(bounces=1, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0)),
                              th4, ph4, int(np.sum(ph4 == ph4[0])), rank, world, dev, coll_dev, stream, 2, collective)
            r4 = build("config-4 fan set-up", make4)
            r4.want_full = True
            s4a = r4.one_pass()                                          # first pass: allocations and first touch of the path chunks (~150 GB), not timed;
            r4.want_full = False                                         # its records (N > 1: the gathered table) are the checked ones
            c4, gerr = None, None
            if rank == 0:
                try:
                    c4 = check_config4(r4, s4a)
                except Exception as e:                                   # noqa: BLE001
                    gerr = e
            agree(gerr is None, collective, coll_dev, "config-4 check", gerr)
            s4, t4, _ = r4.timed(1)
            extras["config4_fan_strong"] = {"value": s4 / t4, "unit": "RK4 ray-steps/s", "seconds_per_pass": t4, "rays": int(r4.n_az * r4.n_theta), "rays_per_gpu": int(len(r4.theta)),
                                            "scaling": "strong", "workload": "GeoAc3D.RngDep 5x5x1400 grid, 1000 az x 1000 incl, bounces=1, CalcAmp=True, azimuth-sharded; one timed pass after an untimed one (allocations warm)",
                                            "checked": c4}
            del r4
        except Exception as e:                                          # the extras never take the line down: rank-local preparation sits inside build(),
            extras["error"] = repr(e)                                   # every other failure point is agreed on by all ranks (agree / FanRun.one_pass): all of them land here together
        try:
            # BASELINE config 5: GeoAcGlobal.RngDep -eig_search to the 64-receiver ring, sharded by receiver
            extras["config5_ring_strong"] = config5_ring(G, args, rank, world, dev, coll_dev, collective, build, agree)
        except Exception as e:                                          # noqa: BLE001
            extras["config5_error"] = repr(e)

    if rank == 0:
        value = total_steps / dt
        rk4_ms, post_ms, rk4_launches = ev["ms_rk4"], ev["ms_post"], ev["epochs"]
        # dominant kernel = k_rk4: algorithmic bytes per launch / average launch duration (HIP events on the kernel's own stream,
        # recorded inside libgeoac_hip around every k_rk4 epoch of the timed passes).  One epoch of this fan is TWO concurrent k_rk4
        # launches of the same duration by construction (the shallow tenth of the rays on k_rk4<EqGlobalPair>, the rest on
        # k_rk4<EqGlobal<true>> with 0.75 x the rows); "launch" below = that pair, bytes and steps are those of both.
        ach_gbs = (B_ALG_PER_STEP * local_steps_per_pass * args.steps) / (rk4_ms * 1e-3) / 1e9 if rk4_ms > 0 else 0.0
        pmc = newest_pmc()
        bps = pmc.get("k_rk4_hbm_bytes_per_ray_step")
        steps_per_launch = local_steps_per_pass * args.steps / max(rk4_launches, 1)
        n_az, n_theta = run.n_az, run.n_theta
        out = {
            "metric": ("RK4 ray-steps/sec, GeoAcGlobal 360x90 ToyAtmo fan; arrivals within 1e-6 of ref" if n_az == 360 else
                       f"RK4 ray-steps/sec, GeoAcGlobal {n_az}x90 ToyAtmo fan ({n_gpus} x the 360x90 fan: weak scaling, 360 azimuths per GPU); arrivals within 1e-6 of ref"),
            "value": value, "unit": "RK4 ray-steps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"GeoAcGlobal -prop ToyAtmo.met, {n_az} az x {n_theta} incl = {n_az * n_theta} rays "
                                   f"(phi step {1.0 / mult:g} deg), bounces=2, CalcAmp=True, WriteRays=False, rng_max=1500",
                       "rays_per_gpu": int(len(run.theta)), "ray_steps_per_pass": int(total_steps // args.steps),
                       "timed_region": "launch angles H2D + RK4 / post-pass / sum kernels + " + ("RCCL all_gather of the arrival tables + every rank's own arrivals D2H" if collective else "arrival records D2H (geoac_fan_run)") + "; atmosphere tables resident",
                       "parallelism": f"azimuth-sharded x{n_gpus}" + (f", {'RCCL' if args.backend == 'nccl' else args.backend} all_gather of arrivals" if collective else "")},
            "parity_gate": gate,
            "launch_only": {"value": local_steps_per_pass * args.steps * (n_gpus if args.scaling == "weak" else 1) / (ev["ms_total"] * 1e-3) if ev["ms_total"] > 0 and world == 1 else None,
                            "ms_per_step": ev["ms_total"] / args.steps, "what": "HIP events around the kernels of a pass on this rank, no copies"},
            "roofline": {"bound": "fp64_valu_latency",
                         "bound_note": "a register-resident FP64 ODE recurrence: the pass lasts as long as the serial integration of its longest ray; "
                                       "`achieved`/`frac` are the SURVEY 8(d) contract figure (144 B per step against HBM peak), `hbm_measured` the counter bytes, `fp64` the issue-side figure",
                         "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS,
                         "traffic": (bps * steps_per_launch) if bps else None, "traffic_unit": "bytes per launch",
                         "traffic_source": pmc.get("_source"), "traffic_stale": pmc.get("_stale"), "build_id": pmc.get("_build_id_now"),
                         "traffic_note": "PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 passes of this command) cannot be collected inside the run: "
                                         "they come from the named profile; traffic_stale = the profiled library is not the one loaded now",
                         "achieved_bytes_per_launch": B_ALG_PER_STEP * steps_per_launch,
                         "kernel": "k_rk4<EqGlobalPair,true,false> || k_rk4<EqGlobal<true>,true,false> (one epoch)", "launches": rk4_launches,
                         "avg_launch_ms": rk4_ms / max(rk4_launches, 1),
                         "alg_bytes_per_step": B_ALG_PER_STEP,
                         "rk4_ms_per_pass": rk4_ms / args.steps, "postpass_ms_per_pass": post_ms / args.steps},
        }
        if bps and rk4_ms > 0:
            mg = bps * local_steps_per_pass * args.steps / (rk4_ms * 1e-3) / 1e9
            out["roofline"]["hbm_measured"] = {"bytes_per_step": bps, "achieved": mg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": mg / HBM_PEAK_GBS,
                                               "source": pmc.get("_source")}
        fps = pmc.get("k_rk4_fp64_flop_per_ray_step")
        if fps and rk4_ms > 0:
            tf = fps * local_steps_per_pass * args.steps / (rk4_ms * 1e-3) / 1e12
            # the honest "how busy is the binding unit" figure beside the contract's HBM roofline
            out["roofline"]["fp64"] = {"flop_per_step": fps, "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": tf / FP64_PEAK_TFLOPS, "source": pmc.get("_source")}
        # issue-side roofline of the dominant kernel: the pass lasts as long as the serial integration of its longest ray, one wave alone on its
        # SIMD issues at most one instruction per 4 cycles - floor = static instructions per step x 4 cycles / 2.4 GHz against the measured time per step
        isa = newest_isa_mix()
        longest = int(run.longest_ray_steps) if getattr(run, "longest_ray_steps", 0) else 0
        if isa and longest and rk4_ms > 0:
            n_ins = isa["in_line_blocks_skipped"]["instructions"] + isa.get("step_size_block_instructions", 0)
            meas_us = rk4_ms / args.steps / longest * 1e3
            out["roofline"]["issue"] = {"instructions_per_step": n_ins, "cycles_per_instruction_floor": 4, "clock_ghz": 2.4, "floor_us_per_step": n_ins * 4 / 2.4e3,
                                        "measured_us_per_step": meas_us, "frac": (n_ins * 4 / 2.4e3) / meas_us, "longest_ray_steps": longest,
                                        "cycles_per_instruction_measured": meas_us * 2.4e3 / n_ins, "source": isa.get("_source"),
                                        "what": "k_rk4<EqGlobalPair>: static instructions of the step loop's common path (tools/isa_mix.py --trace) x 4 cycles against "
                                                "RK4 time per pass / steps of the longest ray"}
        if world > 1:
            out["scaling_note"] = ("weak: per-GPU work fixed (360 azimuths x 90 inclinations per GPU, fan of N x 360 azimuths); the fixed 360x90 fan does NOT strong-scale - "
                                   "one GPU already integrates it in the time of its longest ray (54 130 steps x the step latency), expected ~1.0-1.1x at 8 GPUs; its measured figure "
                                   "is other_scalings.metric_fan_strong" if args.scaling == "weak" else
                                   "strong: the fixed 360x90 fan dealt over N GPUs; a fan lasts as long as its longest ray, so this figure stays near the 1-GPU one by construction")
        if extras:
            out["other_scalings"] = extras
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
