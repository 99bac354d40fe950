"""CPU suite: the N > 1 path (azimuth sharding + gather of arrival records) on world_size-2 and -3 gloo groups.
Rank-count invariance: the gathered table must equal the single-process table byte for byte."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import geoac_amd as G
from geoac_amd.sharding import gather_eigenrays, gather_records, shard_by_azimuth, shard_receivers, shard_sizes


def _fake_records(theta, phi, legs=3, stride=32):
    """deterministic stand-in for the per-ray records (the real ones come from the GPU)"""
    n = len(theta)
    rec = np.zeros((n, legs, stride))
    for l in range(legs):
        rec[:, l, 0] = 1.0
        rec[:, l, 1] = np.floor(1000 * theta + 7 * (phi + 200) + l)
        rec[:, l, 3] = theta * 17.0 + phi * 0.25 + l
        rec[:, l, 12] = np.sin(theta) + np.cos(phi) * (l + 1)
    return rec


def _worker(rank, world, port, n_az, n_theta, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    phi_step = 360.0 / n_az
    th, ph = [], []
    for a in range(n_az):
        for t in range(n_theta):
            th.append(0.5 + 0.5 * t); ph.append(-180.0 + a * phi_step)
    th, ph = np.array(th), np.array(ph)
    thl, phl, idx = shard_by_azimuth(th, ph, n_theta, rank, world)
    rec_local = torch.from_numpy(_fake_records(thl, phl))
    full = gather_records(rec_local, n_az, n_theta)
    steps = torch.tensor([int(rec_local[:, :, 1].sum().item())])
    dist.all_reduce(steps)
    want = _fake_records(th, ph)
    ok = np.array_equal(full.numpy(), want) and int(steps.item()) == int(want[:, :, 1].sum())
    q.put((rank, ok, len(thl)))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world,n_az", [(2, 8), (3, 8), (2, 5)])
def test_gather_is_rank_count_invariant(world, n_az):
    n_theta = 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_az, n_theta, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert sorted(n for _, _, n in res) == sorted(shard_sizes(n_az, n_theta, world))


def test_shard_by_azimuth_covers_fan_once():
    th, ph = G.fan_enumerate(phi_min=-180.0, phi_max=179.0, phi_step=1.0) if os.path.exists(G.library_path()) else (None, None)
    if th is None:
        pytest.skip("library not built")
    seen = np.zeros(len(th), dtype=int)
    for r in range(8):
        _, _, idx = shard_by_azimuth(th, ph, 90, r, 8)
        seen[idx] += 1
    assert (seen == 1).all()


def _fake_eigenrays(rcvr_global):
    """deterministic stand-in: receiver g has (g % 3) eigenrays"""
    rows = []
    for local, g in enumerate(rcvr_global):
        for k in range(int(g) % 3):
            row = np.zeros(16); row[0] = local; row[1] = k; row[3] = 5.0 + g + 0.25 * k; row[5] = 800.0 + g
            rows.append(row)
    return np.array(rows).reshape(-1, 16)


def _eig_worker(rank, world, port, n_rcvr, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_receivers(n_rcvr, rank, world)
    full = gather_eigenrays(torch.from_numpy(_fake_eigenrays(mine)), mine)
    want = _fake_eigenrays(np.arange(n_rcvr))
    want[:, 0] = [g for g in range(n_rcvr) for _ in range(g % 3)]
    q.put((rank, np.array_equal(full.numpy(), want)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rcvr", [(2, 7), (3, 8)])
def test_receiver_sharding_gathers_every_eigenray_once(world, n_rcvr):
    """config 5 shards by receiver: the gathered eigenray table must not depend on the rank count"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eig_worker, args=(r, world, port, n_rcvr, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_gather_eigenrays_without_a_process_group_is_the_local_table():
    """N = 1 (bench.py's config-5 leg on one GPU, no process group): the local table with global receiver indices, ordered"""
    mine = np.array([5, 2, 7])
    full = gather_eigenrays(torch.from_numpy(_fake_eigenrays(mine)), mine).numpy()
    assert [int(x) for x in full[:, 0]] == [2, 2, 5, 5, 7]          # receiver g has g % 3 eigenrays
    assert [int(x) for x in full[:, 1]] == [0, 1, 0, 1, 0]


def test_bench_refuses_a_rank_count_that_is_not_gpus(tmp_path):
    """bench.py under a launcher whose world size differs from --gpus must not print a line that claims N GPUs"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b"--gpus 8 but the launcher started 1" in r.stderr and b"{" not in r.stdout
