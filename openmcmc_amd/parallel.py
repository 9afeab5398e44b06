"""Multi-GPU layout of the path: chains are independent, so ranks share nothing while sampling.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Chains are split into
contiguous blocks; the random streams are keyed by the GLOBAL chain id (chain_id_offset), so the
union of all ranks' chains is identical to a single-GPU run of all chains.  The only collective is
the gather of the per-rank stores at the end of the run.
"""

import numpy as np


def shard_chains(n_chains_total, world_size, rank):
    """(n_local, chain_id_offset) of `rank`: contiguous blocks, remainder to the low ranks."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, rem = divmod(int(n_chains_total), int(world_size))
    n_local = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return n_local, offset


def store_to_reference_layout(key, array):
    """(n_iter, C, size) device layout -> the reference's per-chain layout (C, size, n_iter);
    log_post (n_iter, C) -> (C, n_iter, 1)."""
    if key == "log_post":
        return np.transpose(array, (1, 0))[:, :, None]
    return np.transpose(array, (1, 2, 0))


def gather_store(store, dst=0, group=None):
    """Gather {key: (n_iter, C_local, ...)} tensors from all ranks on `dst` (chains concatenated in
    rank order = global chain order) and return host arrays in the reference layout; None on the
    other ranks.  Ranks may hold different chain counts.  Works on any backend (RCCL on GPUs, gloo
    on CPU tensors in the tests)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return {k: store_to_reference_layout(k, t.detach().cpu().numpy()) for k, t in store.items()}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [None] * world
    first = store[sorted(store)[0]]
    dist.all_gather_object(counts, int(first.shape[1]), group=group)
    result = {} if rank == dst else None
    for key in sorted(store):
        t = store[key].contiguous()
        bucket = None
        if rank == dst:
            bucket = [t.new_empty((t.shape[0], counts[r]) + tuple(t.shape[2:])) for r in range(world)]
        if len(set(counts)) == 1:
            dist.gather(t, bucket, dst=dst, group=group)
        else:  # uneven shards: point-to-point into the root
            if rank == dst:
                for r in range(world):
                    if r == dst:
                        bucket[r].copy_(t)
                    else:
                        dist.recv(bucket[r], src=r, group=group)
            else:
                dist.send(t, dst=dst, group=group)
        if rank == dst:
            result[key] = store_to_reference_layout(key, torch.cat(bucket, dim=1).cpu().numpy())
    return result
