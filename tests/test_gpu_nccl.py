"""The RCCL path of the N > 1 flow, executed on the one GPU of the test box: bench.py under torch.distributed.run with ONE rank and
--force-collective - init_process_group("nccl", device_id=...), the ok-flag all_reduce, geoac_fan_copy_records_dev on torch's current stream,
all_gather_into_tensor of the record tables and all_reduce of the step counts on device tensors (geoac_amd/sharding.py), barriers - and the
gathered table judged by bench.py's own parity gate against the reference-made fixture of the whole metric fan (every ray, counts exact,
values to 1e-6).  The launcher is started as a child process: nothing that has touched the GPU is replaced by another program."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_bench_flow_over_rccl_with_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--force-collective", "--backend", "nccl"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    print("RCCL world-1 run:", d["value"], d["unit"], "|", d["config"]["parallelism"], "|", d["parity_gate"]["status"], d["parity_gate"]["max_rel_err"])
    assert d["parity_gate"]["status"] == "pass" and d["parity_gate"]["rays_checked"] == 32400      # the GATHERED table against the reference's fan
    assert "RCCL all_gather" in d["config"]["parallelism"] and "RCCL all_gather" in d["config"]["timed_region"]
    assert d["config"]["ray_steps_per_pass"] == 874273730


def test_bench_gpus_2_starts_its_two_ranks_itself():
    """`python bench.py --gpus 2` with NO launcher around it: bench.py must start the two ranks (child torch.distributed.run, before it touches a
    GPU), relay their JSON line and merge the host-side cpu_baseline.  gloo: both ranks share the one GPU of this box (the RCCL transport is
    covered above); weak scaling, so the fan has 720 azimuths and the parity gate checks every second one against the reference's fan."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "1", "--no-extras"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line"
    d = json.loads(lines[0])
    print("bench --gpus 2 (self-started ranks):", d["value"], d["unit"], "|", d["config"]["parallelism"], "|", d["launcher"])
    assert d["n_gpus"] == 2 and d["config"]["rays_per_gpu"] == 32400
    assert d["parity_gate"]["status"] == "pass" and d["parity_gate"]["rays_checked"] == 32400
    assert d["config"]["ray_steps_per_pass"] > 2 * 870000000
    assert d["cpu_baseline"]["value"] > 1e5 and d["cpu_baseline"]["cores"] == 1            # measured by the parent, merged into the ranks' line
    assert "roofline" in d and d["roofline"]["frac"] > 0
