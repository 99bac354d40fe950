"""Worker of tests/test_gpu_eig_ring.py::test_config5_sharded_by_receiver_over_two_ranks: one rank of a world_size-N run (gloo; the ranks
share the GPU of the test box): receivers sharded round robin -> a REAL geoac_eig_search on the GPU -> gather_eigenrays -> rank 0 saves
the gathered table.  argv: out.npz ring positions...   Run under torch.distributed.run."""
import os
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import geoac_amd as G  # noqa: E402
import rngdep_data as RD  # noqa: E402
from geoac_amd.sharding import gather_eigenrays, shard_receivers  # noqa: E402
from parity import ring_receivers  # noqa: E402


def main():
    out, pos = sys.argv[1], [int(a) for a in sys.argv[2:]]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rcv = ring_receivers()[pos]
    mine = shard_receivers(len(rcv), rank, world)
    with tempfile.TemporaryDirectory() as td:
        ctx = G.FanContext(G.EQ_GLOBAL_RNGDEP, device=0)
        ctx.load_grid(*RD.write_grid_global(td, short_paths=False))
        ctx.set_params(src=(0.0, 31.0, 0.0))
        res = ctx.eig_search(rcv[mine], bnc_min=0, bnc_max=2, verbose=False)
    full = gather_eigenrays(torch.from_numpy(res["eig"]), mine)
    shares = [None] * world
    dist.all_gather_object(shares, mine.tolist())
    if rank == 0:
        np.savez(out, eig=full.numpy(), **{f"mine{r}": np.array(shares[r]) for r in range(world)})
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
