"""world_size-2 gloo rehearsal of the multi-GPU path on CPU: chain sharding and the single
gather of the stores (the only collective of the path)."""

import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openmcmc_amd.parallel import gather_store, shard_chains


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local, offset = shard_chains(total, world, rank)
    n_iter, size = 3, 4
    # a store whose entries encode (iteration, GLOBAL chain id, element)
    it = torch.arange(n_iter, dtype=torch.float64)[:, None, None]
    ch = (offset + torch.arange(n_local, dtype=torch.float64))[None, :, None]
    el = torch.arange(size, dtype=torch.float64)[None, None, :]
    store = {"b": 10000 * it + 100 * ch + el, "log_post": (10 * it + ch)[:, :, 0] * torch.ones(n_iter, n_local, dtype=torch.float64)}
    out = gather_store(store, dst=0)
    if rank == 0:
        q.put({k: v for k, v in out.items()})
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _run(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return out


def test_gather_even_shards():
    out = _run(6)
    assert out["b"].shape == (6, 4, 3) and out["log_post"].shape == (6, 3, 1)
    for c in range(6):
        for i in range(3):
            assert np.array_equal(out["b"][c, :, i], 10000 * i + 100 * c + np.arange(4.0))
            assert out["log_post"][c, i, 0] == 10 * i + c


def test_gather_uneven_shards():
    out = _run(5)  # ranks hold 3 and 2 chains
    assert out["b"].shape == (5, 4, 3)
    for c in range(5):
        assert np.array_equal(out["b"][c, :, 0], 100 * c + np.arange(4.0))


def _stream_worker(rank, world, port, total, every, q):
    """The streaming form of the same collective: a ring store's halves go to the root one by one (parallel.GatherSink)."""
    import types

    from openmcmc_amd.parallel import GatherSink

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local, offset = shard_chains(total, world, rank)
    n_iter, size, half = 7, 4, 2
    it = torch.arange(n_iter, dtype=torch.float64)[:, None, None]
    ch = (offset + torch.arange(n_local, dtype=torch.float64))[None, :, None]
    el = torch.arange(size, dtype=torch.float64)[None, None, :]
    full = {"b": 10000 * it + 100 * ch + el, "log_post": ((10 * it + ch)[:, :, 0]).contiguous()}
    sink = GatherSink(dst=0, every=every)
    sink.bind(types.SimpleNamespace(n_iter=n_iter, n_chains=n_local, seed=0, chain_id_offset=offset, engine=None), None)
    for it0 in range(0, n_iter, half):           # what the ring's drain hands over, chunk by chunk
        it1 = min(n_iter, it0 + half)
        for key, t in full.items():
            sink(key, it0, it1, t[it0:it1])
    out = sink.result()
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _run_stream(total, every):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, total, every, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return out


def test_streaming_gather_places_every_chunk():
    for every in (1, 3):
        out = _run_stream(5, every)  # ranks hold 3 and 2 chains; 7 iterations in halves of 2
        kept = list(range(0, 7, every))
        assert out["b"].shape == (5, 4, len(kept)) and out["log_post"].shape == (5, len(kept), 1)
        for c in range(5):
            for j, i in enumerate(kept):
                assert np.array_equal(out["b"][c, :, j], 10000 * i + 100 * c + np.arange(4.0))
                assert out["log_post"][c, j, 0] == 10 * i + c
