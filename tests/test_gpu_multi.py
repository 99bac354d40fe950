"""The multi-GPU paths with REAL records on the one GPU of the test box:
  * torch.distributed (gloo, two and three processes sharing the GPU): azimuth shard -> fan launch -> gather == the single-process table,
    byte for byte (the N > 1 flow of bench.py / geoac_amd/sharding.py; with RCCL only the transport differs);
  * geoac_pool (include/geoac_multi.h): two contexts on the same device, groups from the shared queue == one context;
  * the -prop drivers with GEOAC_DEVICES=0,0 write the same files as with one device, and GEOAC_STATS the run summary."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world", [2, 3])
def test_shard_launch_gather_equals_single_process(world, tmp_path):
    import geoac_amd as G
    out = str(tmp_path / "gathered.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(H.ROOT, "tests", "mp_shard_worker.py"), out]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    g = np.load(out)
    ctx = G.FanContext(G.EQ_GLOBAL, device=0)
    ctx.load_met(H.TOYATMO)
    ctx.set_params(bounces=2, calc_amp=1, mode=0)
    rec, steps = ctx.run(g["theta"], g["phi"])
    assert int(g["steps"]) == steps
    assert np.array_equal(g["rec"], rec)


def test_pool_of_two_contexts_equals_one_context(tmp_path):
    import geoac_amd as G
    import rngdep_data as RD
    # stratified spherical set: 24 azimuths x 30 inclinations in groups of 60 rays
    th, ph = G.fan_enumerate(theta_min=1.0, theta_max=44.5, theta_step=1.5, phi_min=-180.0, phi_max=165.0, phi_step=15.0)
    one = G.FanContext(G.EQ_GLOBAL, device=0); one.load_met(H.TOYATMO); one.set_params(bounces=1, calc_amp=1, mode=0)
    want, steps = one.run(th, ph)
    pool = G.FanPool(G.EQ_GLOBAL, [0, 0]); pool.load_met(H.TOYATMO); pool.set_params(bounces=1, calc_amp=1, mode=0)
    got, st = pool.run(th, ph, rays_per_group=60)
    sh = pool.shares()
    print("shares", sh)
    assert st == steps and np.array_equal(got, want)
    assert sum(sh["rays"]) == len(th) and sum(sh["groups"]) == 12 and min(sh["groups"]) >= 1
    got2, st2 = pool.run(th, ph)                                   # automatic group size
    assert st2 == steps and np.array_equal(got2, want)
    # range-dependent Cartesian set through the pool (grid table built on each context's device)
    grid = RD.write_grid(str(tmp_path), short_paths=False)
    th, ph = G.fan_enumerate(theta_min=3.0, theta_max=39.0, theta_step=4.0, phi_min=-90.0, phi_max=90.0, phi_step=45.0)
    one = G.FanContext(G.EQ_3D_RNGDEP, device=0); one.load_grid(*grid); one.set_params(bounces=1, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
    want, steps = one.run(th, ph)
    pool = G.FanPool(G.EQ_3D_RNGDEP, [0, 0]); pool.load_grid(*grid); pool.set_params(bounces=1, calc_amp=1, mode=0, src=(0.0, 0.0, 0.0))
    got, st = pool.run(th, ph, rays_per_group=10)
    assert st == steps and np.array_equal(got, want)
    with pytest.raises(G.GeoAcError):
        pool.set_params(mode=1)                                    # sample capture is per context


def test_prop_driver_on_two_devices_writes_the_same_files(tmp_path):
    import shutil
    exe = os.path.join(H.ROOT, "geoac_amd", "bin", "GeoAcGlobal")
    args = ["-prop", "ToyAtmo.met", "theta_min=3", "theta_max=43", "theta_step=4", "phi_min=-180", "phi_max=150", "phi_step=30",
            "bounces=1", "WriteRays=False"]
    outs = []
    for tag, devs in (("one", None), ("two", "0,0")):
        d = tmp_path / tag; d.mkdir()
        shutil.copy(H.TOYATMO, d / "ToyAtmo.met")
        gpu = ["gpu_stats=" + str(d / "stats.json")] + (["gpu_devices=" + devs] if devs else [])        # arguments of the GPU build, not environment
        r = subprocess.run([exe] + args + gpu, cwd=d, env=dict(os.environ), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode()
        outs.append((open(d / "ToyAtmo_results.dat").read(), json.load(open(d / "stats.json"))))
    assert outs[0][0] == outs[1][0] and len(outs[0][0].split("\n")) > 100
    s1, s2 = outs[0][1], outs[1][1]
    assert s1["rk4_ray_steps"] == s2["rk4_ray_steps"] > 0 and s1["rays"] == s2["rays"] == 12 * 11
    assert s2["devices"] == [0, 0] and sum(p["rays"] for p in s2["per_device"]) == s2["rays"]
