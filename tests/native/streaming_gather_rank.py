"""One rank of tests/test_store_gpu.py::test_streaming_gather_two_ranks_share_the_gpu (gloo; RANK / WORLD_SIZE / MASTER_* from
the environment, OUT = where rank 0 leaves the gathered store)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist

    from openmcmc_amd.parallel import GatherSink, shard_chains
    from test_mcmc_api_gpu import build

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    G = np.load(os.path.join(ROOT, "tests", "golden", "gmrf_chain.npz"))
    C, offset = shard_chains(5, world, rank)  # 3 + 2 chains: uneven shards
    sink, thin = GatherSink(dst=0), GatherSink(dst=0, every=3)

    def both(key, it0, it1, block):
        sink(key, it0, it1, block)
        if key == "b":
            thin(key, it0, it1, block)

    both.bind = lambda mcmc, stream: (sink.bind(mcmc, stream), thin.bind(mcmc, stream))
    M, _ = build(G, "sparse_", True, C, fuse=True, n_burn=2, n_iter=11, seed=4, chain_id_offset=offset, store_ring=4, sink=both)
    M.run_mcmc()
    res, res_thin = sink.result(), thin.result()
    if rank == 0:
        np.savez(os.environ["OUT"], thin_b=res_thin["b"], **res)
    else:
        assert res is None
    M.engine.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
