"""Reversible-jump move bookkeeping on the GPU (INT / index path): 10 000-step traces of the
reference (tests/golden/rj_moves.npz) replayed with its uniforms and deletion indices injected --
move type, probabilities and indices bit-exact; the in-kernel bounded integers are unbiased."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_engine(C, **kw):
    from openmcmc_amd.engine import Engine

    return Engine(C, **kw)


@pytest.mark.parametrize("case", range(8))
def test_rj_trace_bit_exact(golden, case):
    import torch

    G = golden("rj_moves")
    n_max, q, _ = G["cases"][case]
    rows = G[f"case{case}"]
    C = rows.shape[0]  # one "chain" per recorded step: every visited state is checked in one launch
    eng = make_engine(C)
    n = torch.as_tensor(rows[:, 0].astype(np.int64), device="cuda")
    u = eng.to_device(np.where(rows[:, 1] >= 0, rows[:, 1], 0.5))
    idx = torch.as_tensor(rows[:, 5].astype(np.int64), device="cuda")
    birth, pb, pd, dele = eng.rj_move(n, int(n_max), float(q), u=u, idx=idx)
    eng.check_status()
    assert np.array_equal(birth.cpu().numpy(), rows[:, 2].astype(np.int32))
    assert np.array_equal(pb.cpu().numpy(), rows[:, 3]) and np.array_equal(pd.cpu().numpy(), rows[:, 4])
    assert np.array_equal(dele.cpu().numpy(), rows[:, 5].astype(np.int64))
    eng.close()


def test_rj_in_kernel_draws_and_errors():
    import torch

    C = 60000
    eng = make_engine(C, seed=4)
    n = torch.full((C,), 7, dtype=torch.int64, device="cuda")
    birth, pb, pd, dele = eng.rj_move(n, 20, 0.3, draw_index=5)
    eng.check_status()
    b = birth.cpu().numpy()
    assert abs(b.mean() - 0.3) < 0.01
    d = dele.cpu().numpy()
    assert np.all(d[b == 1] == -1)
    counts = np.bincount(d[b == 0], minlength=7)
    assert counts.shape[0] == 7 and np.all(np.abs(counts / counts.sum() - 1 / 7) < 0.01)
    n[123] = 0  # the reference raises ValueError for n == 0 (reversible_jump.py:330-331)
    eng.rj_move(n, 20, 0.3)
    with pytest.raises(np.linalg.LinAlgError, match="chain 123"):
        eng.check_status()
    eng.close()
