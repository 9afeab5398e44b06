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
    """(n_iter, C, size) device layout -> the reference's per-chain layout (C, size, n_iter); (n_iter, C, rows, cols) of a
    matrix-valued variable-size parameter -> (C, rows, cols, n_iter) (sampler.py:81-82); log_post (n_iter, C) -> (C, n_iter, 1)."""
    if key == "log_post":
        return np.transpose(array, (1, 0))[:, :, None]
    return np.moveaxis(array, 0, -1)  # (n_iter, C, ...) -> (C, ..., n_iter); a tuple max_variable_size keeps its two axes


def make_communicator(engine, group=None):
    """The library's own RCCL communicator over the ranks of `group` (omc_comm_create): rank 0 makes the bootstrap
    id and torch.distributed's object broadcast ships it (any other channel would do for a non-Python caller)."""
    import torch.distributed as dist

    from openmcmc_amd.engine import new_unique_id

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [new_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return engine.communicator(world, rank, box[0])


def gather_chains(t, chain_dim, dst=0, group=None, comm=None):
    """Gather a per-rank tensor whose dimension `chain_dim` indexes this rank's chains: on `dst` the tensor with the
    chains of all ranks concatenated in rank order (= global chain order), None elsewhere.  Ranks may hold
    different chain counts.  With `comm` (an engine.Communicator) the transfer is the library's omc_gather_samples
    on RCCL; without, torch.distributed's gather / send / recv on whatever backend the group has (gloo on CPU
    tensors in the tests)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [None] * world
    dist.all_gather_object(counts, int(t.shape[chain_dim]), group=group)
    if comm is not None:
        # (outer..., C, inner...) -> (n_outer, C, row): the dimensions in front of the chains are the outer index
        lead = t.shape[:chain_dim]
        blk = t.contiguous().reshape((int(np.prod(lead)) if lead else 1, t.shape[chain_dim]) + tuple(t.shape[chain_dim + 1:]))
        out = comm.gather(blk, counts, root=dst)
        return out.reshape(tuple(lead) + (sum(counts),) + tuple(t.shape[chain_dim + 1:])) if rank == dst else None
    t = t.contiguous()
    shape = list(t.shape)

    def like(c):
        shape[chain_dim] = c
        return t.new_empty(shape)

    bucket = [like(counts[r]) for r in range(world)] if rank == dst else None
    if len(set(counts)) == 1:
        dist.gather(t, bucket, dst=dst, group=group)
    elif rank == dst:  # uneven shards: point to point into the root
        for r in range(world):
            if r == dst:
                bucket[r].copy_(t)
            else:
                dist.recv(bucket[r], src=r, group=group)
    else:
        dist.send(t, dst=dst, group=group)
    return torch.cat(bucket, dim=chain_dim) if rank == dst else None


def gather_store(store, dst=0, group=None, comm=None):
    """Gather {key: (n_iter, C_local, ...)} tensors from all ranks on `dst` (chains concatenated in
    rank order = global chain order) and return host arrays in the reference layout; None on the
    other ranks.  Ranks may hold different chain counts.  With `comm` (make_communicator) the transfers are the
    library's own RCCL gather; without, torch.distributed's on the group's backend (gloo on CPU tensors in the
    tests)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return {k: store_to_reference_layout(k, t.detach().cpu().numpy()) for k, t in store.items()}
    rank = dist.get_rank(group)
    result = {} if rank == dst else None
    for key in sorted(store):
        full = gather_chains(store[key], 1, dst=dst, group=group, comm=comm)
        if rank == dst:
            result[key] = store_to_reference_layout(key, full.cpu().numpy())
    return result
