"""omc_gather_samples (RCCL) through the C ABI.  One GPU box = one rank: the single-rank communicator exercises the
bootstrap (unique id, ncclCommInitRank), the argument checks and the root's own-block placement; the layout with several
ranks and uneven shards is held by tests/test_distributed_cpu.py (gloo) against the same `gather_chains` contract."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_gather_places_own_block():
    from openmcmc_amd.engine import Engine, new_unique_id

    eng = Engine(5, seed=1)
    uid = new_unique_id()
    assert len(uid) == 128
    comm = eng.communicator(1, 0, uid)
    rng = np.random.default_rng(0)
    for shape in [(3, 5, 7), (1, 5, 11), (4, 5)]:
        host = rng.standard_normal(shape)
        out = comm.gather(eng.to_device(host), [5], root=0)
        eng.synchronize()
        assert out.shape == shape
        assert np.array_equal(out.cpu().numpy(), host)
    with pytest.raises(ValueError):
        comm.gather(eng.to_device(rng.standard_normal((3, 4, 2))), [5])
    comm.close()
    eng.close()


def test_mcmc_gather_with_library_communicator_matches_collect():
    """MCMC.gather(comm=...) on one rank returns what collect() returns (reference layout)."""
    import torch.distributed as dist

    from openmcmc_amd.engine import Engine, new_unique_id
    from openmcmc_amd.parallel import gather_chains

    eng = Engine(4, seed=3)
    comm = eng.communicator(1, 0, new_unique_id())
    t = eng.to_device(np.arange(2 * 4 * 3, dtype=float).reshape(2, 4, 3))
    assert not dist.is_initialized()
    assert gather_chains(t, 1, comm=comm) is t  # no process group: nothing to gather
    out = comm.gather(t, [4])
    eng.synchronize()
    assert np.array_equal(out.cpu().numpy(), t.cpu().numpy())
    comm.close()
    eng.close()


@pytest.mark.parametrize("world,counts", [(2, [4, 4]), (2, [5, 2]), (3, [3, 1, 4]), (3, [0, 6, 2]), (8, [2] * 8),
                                          (8, [3, 1, 4, 1, 5, 2, 6, 5])])
@pytest.mark.parametrize("root", [0, 1])
def test_root_side_placement_for_several_ranks(world, counts, root):
    """omc_gather_samples for W > 1 as the root sees it -- slab loop under the staging budget, staging offsets in rank order,
    direct placement for one outer index, the interleave kernel -- with the peers' blocks supplied from this GPU
    (omc_gather_samples_local: the same code path, device-to-device copies where the RCCL receives stand) against
    numpy.concatenate.  RCCL itself needs one GPU per rank and is not reachable from a one-GPU box."""
    from openmcmc_amd.engine import Engine, gather_local

    eng = Engine(max(1, max(counts)), seed=1)
    rng = np.random.default_rng(world * 10 + root)
    for n_outer, tail in [(1, (7,)), (5, (7,)), (5, ()), (3, (2, 3)), (17, (5,))]:
        host = [rng.standard_normal((n_outer, c) + tail) for c in counts]
        want = np.concatenate(host, axis=1)
        blocks = [eng.to_device(h) if h.size else eng.empty(*h.shape) for h in host]
        row = int(np.prod(tail)) if tail else 1
        per_outer = (sum(counts) - counts[root]) * row * 8
        # default budget (one slab), one outer index per slab, a budget that does not divide the outer count, and less than one
        for limit in (0, per_outer, 2 * per_outer + 8, 3 * per_outer - 8, 1):
            out = gather_local(eng, blocks, root=root, staging_limit_bytes=limit)
            eng.synchronize()
            assert out.shape == want.shape
            assert np.array_equal(out.cpu().numpy(), want), (n_outer, tail, limit)
    eng.close()
