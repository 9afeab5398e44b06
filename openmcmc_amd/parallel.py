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


def gather_chains(t, chain_dim, dst=0, group=None, comm=None, counts=None):
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
    if counts is None:  # (a caller that gathers many blocks of the same sharding passes the counts it already has)
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


class GatherSink:
    """Streaming form of the run's one collective, for MCMC(store_ring=R, sink=GatherSink(...)): every half of the ring the
    drain stream empties goes straight to rank `dst` (chains of all ranks in global chain order) while the chains keep
    sampling, so no rank ever holds more than R iteration slabs on its GPU and the root never receives the whole store in
    one burst at the end.  `result()` (after run_mcmc) -> the dict of MCMC.gather() on dst, None elsewhere.

    comm="library": the transfers are omc_gather_samples on a second context bound to the drain stream (RCCL, every peer
    point to point into the root); None: torch.distributed's gather on the group's backend (gloo ranks stage through the host).
    every=k keeps iterations 0, k, 2k, ... only (a thinned streaming gather)."""

    def __init__(self, dst=0, group=None, comm=None, every=1):
        self.dst, self.group, self.comm_kind, self.every = int(dst), group, comm, max(1, int(every))
        self.comm, self.host, self.counts, self._eng2, self.n_iter = None, {}, None, None, None

    def bind(self, mcmc, stream):
        """called once by the ring's drain (under the drain stream) before the first chunk"""
        import torch
        import torch.distributed as dist

        self.n_iter = mcmc.n_iter
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        self.rank = dist.get_rank(self.group) if self.active else 0
        self.cpu_staged = self.active and dist.get_backend(self.group) != "nccl"
        if self.active:
            counts = [None] * dist.get_world_size(self.group)
            dist.all_gather_object(counts, int(mcmc.n_chains), group=self.group)
            self.counts = counts
            if self.comm_kind == "library" and not self.cpu_staged:
                from openmcmc_amd.engine import Engine

                with torch.cuda.stream(stream):  # a context of its own on the drain stream: its collective must not queue behind the sweeps
                    self._eng2 = Engine(mcmc.n_chains, seed=mcmc.seed, device=mcmc.engine.device_index, chain_id_offset=mcmc.chain_id_offset)
                    self.comm = make_communicator(self._eng2, self.group)

    def __call__(self, key, it0, it1, block):
        import torch

        first = (-it0) % self.every  # first iteration of the chunk that is a multiple of `every`
        if first >= it1 - it0:
            return
        part = block[first:: self.every]
        j0 = (it0 + first) // self.every
        if self.cpu_staged:
            part = part.cpu()
        full = gather_chains(part, 1, dst=self.dst, group=self.group, comm=self.comm, counts=self.counts) if self.active else part
        if self.rank != self.dst:
            return
        if key not in self.host:
            kept = (self.n_iter + self.every - 1) // self.every
            self.host[key] = torch.empty((kept,) + tuple(full.shape[1:]), dtype=full.dtype, pin_memory=torch.cuda.is_available())
        self.host[key][j0: j0 + full.shape[0]].copy_(full, non_blocking=True)

    def result(self):
        import torch

        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        if self._eng2 is not None:
            self._eng2.close()
            self._eng2 = None
        if self.rank != self.dst:
            return None
        return {key: store_to_reference_layout(key, t.numpy()) for key, t in self.host.items()}
