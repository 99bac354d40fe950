"""Worker of tests/test_gpu_multi.py: one rank of a world_size-N run on ONE GPU (gloo): azimuth shard -> real fan launch on the GPU ->
gather of the record tables -> rank 0 saves the whole fan's table.  Run under torch.distributed.run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import geoac_amd as G  # noqa: E402
import harness as H  # noqa: E402
from geoac_amd.sharding import gather_records, shard_by_azimuth  # noqa: E402


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    th, ph = G.fan_enumerate(theta_min=2.0, theta_max=44.0, theta_step=3.0, phi_min=-180.0, phi_max=150.0, phi_step=30.0)
    n_theta = int(np.sum(ph == ph[0]))
    thl, phl, _ = shard_by_azimuth(th, ph, n_theta, rank, world)
    ctx = G.FanContext(G.EQ_GLOBAL, device=0)
    ctx.load_met(H.TOYATMO)
    ctx.set_params(bounces=2, calc_amp=1, mode=0)
    rec, steps = ctx.run(thl, phl)
    full = gather_records(torch.from_numpy(rec), len(th) // n_theta, n_theta)
    st = torch.tensor([steps], dtype=torch.int64)
    dist.all_reduce(st)
    if rank == 0:
        np.savez(out, rec=full.numpy(), steps=int(st.item()), theta=th, phi=ph)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
