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
