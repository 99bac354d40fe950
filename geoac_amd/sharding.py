"""Azimuth sharding of a launch-angle fan over ranks (one process per GPU) and the gather of arrival records.

The fan shards with no data-path exchange (rays are independent); the only collective is the gather of the
fixed-stride record tables at the end (RCCL over xGMI when the tensors live on GPUs, gloo on CPU for tests)."""
import numpy as np


def shard_by_azimuth(theta, phi, n_theta, rank, world):
    """rank r takes azimuth indices r, r+world, ... of a phi-major fan with n_theta inclinations per azimuth.
    Returns (theta_local, phi_local, global_ray_index_local)."""
    n = len(theta)
    assert n % n_theta == 0, "fan is not phi-major with a fixed inclination count"
    n_az = n // n_theta
    az = np.arange(rank, n_az, world)
    idx = (az[:, None] * n_theta + np.arange(n_theta)[None, :]).reshape(-1)
    return theta[idx], phi[idx], idx


def shard_sizes(n_az, n_theta, world):
    return [len(range(r, n_az, world)) * n_theta for r in range(world)]


def gather_records(rec_local, n_az, n_theta, group=None):
    """all-gather per-rank record tables [n_local][legs][stride] (torch tensors, same device on every rank) and
    return the table of the whole fan in the reference's ray order.  Ranks may hold different ray counts
    (n_az not divisible by world): tables are padded to the largest shard for the collective."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_az, n_theta, world)
    legs, stride = rec_local.shape[1], rec_local.shape[2]
    nmax = max(sizes)
    pad = torch.zeros((nmax, legs, stride), dtype=rec_local.dtype, device=rec_local.device)
    pad[: rec_local.shape[0]] = rec_local
    out = torch.empty((world * nmax, legs, stride), dtype=rec_local.dtype, device=rec_local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.view(world, nmax, legs, stride)
    full = torch.empty((n_az * n_theta, legs, stride), dtype=rec_local.dtype, device=rec_local.device)
    for r in range(world):
        az = torch.arange(r, n_az, world, device=rec_local.device)
        idx = (az[:, None] * n_theta + torch.arange(n_theta, device=rec_local.device)[None, :]).reshape(-1)
        full[idx] = out[r, : sizes[r]]
    return full


def shard_receivers(n_rcvr, rank, world):
    """eigenray searches shard by receiver: rank r takes receivers r, r+world, ...; returns their indices"""
    return np.arange(rank, n_rcvr, world)


def gather_eigenrays(eig_local, rcvr_index_local, group=None):
    """all-gather per-rank eigenray tables [n_local][EIG_STRIDE] (torch tensors; counts differ per rank) and return one table
    ordered by (global receiver index, eigenray number).  Column 0 of the local tables holds the LOCAL receiver index of
    geoac_eig_search; `rcvr_index_local` (the array shard_receivers returned) maps it back to the global one."""
    import torch
    import torch.distributed as dist
    dev = eig_local.device
    eig = eig_local.clone()
    if eig.shape[0]:
        eig[:, 0] = torch.as_tensor(np.asarray(rcvr_index_local), dtype=eig.dtype, device=dev)[eig[:, 0].long()]
    if not (dist.is_available() and dist.is_initialized()):          # one process, no group: the local table is the whole one
        if eig.shape[0]:
            eig = eig[torch.argsort(eig[:, 0] * 1e6 + eig[:, 1])]
        return eig
    world = dist.get_world_size(group)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([eig.shape[0]], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(max(counts), 1)
    pad = torch.zeros((nmax, eig.shape[1]), dtype=eig.dtype, device=dev)
    pad[: eig.shape[0]] = eig
    out = torch.empty((world * nmax, eig.shape[1]), dtype=eig.dtype, device=dev)
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.view(world, nmax, eig.shape[1])
    full = torch.cat([out[r, : counts[r]] for r in range(world)], dim=0)
    if full.shape[0]:
        key = full[:, 0] * 1e6 + full[:, 1]
        full = full[torch.argsort(key)]
    return full
