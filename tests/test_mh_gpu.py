"""ManifoldMALA / RandomWalk steps on a dense Gaussian target through the C ABI: 40-step traces of
the reference (tests/golden/mala.npz) with its recorded z and u injected.  Accept flags and counters
are integer state: compared bit-exact.  Plus acceptance statistics with in-kernel draws."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def make_engine(C, **kw):
    from openmcmc_amd.engine import Engine

    return Engine(C, **kw)


@pytest.mark.parametrize("d", [1, 5, 32])
@pytest.mark.parametrize("kind", ["mala", "rw"])
def test_mh_trace_golden(golden, d, kind):
    import torch

    G = golden("mala")
    k = f"d{d}_"
    C = 3
    eng = make_engine(C)
    Q = eng.to_device(G[k + "Q"])
    step = float(G[k + kind + "_step"])
    x = eng.to_device(np.tile(G[k + "x0"], (C, 1)))
    acc = torch.zeros(C, dtype=torch.int64, device="cuda")
    prop = torch.zeros(C, dtype=torch.int64, device="cuda")
    if kind == "mala":
        L, sl = eng.dense_cholesky(Q, 1.0 / step**2)
    else:
        L, sl = eng.dense_cholesky(Q, 1.0)
    zs, us = G[k + kind + "_z"], G[k + kind + "_u"]
    flags = []
    for i in range(zs.shape[0]):
        before = acc.clone()
        z = eng.to_device(np.tile(zs[i], (C, 1)))
        u = eng.full((C,), us[i])
        if kind == "mala":
            eng.mala_step(Q, None, L, sl, step, x, z=z, u=u, accept_count=acc, proposal_count=prop)
        else:
            eng.rw_step(None, L, sl, step, x, z=z, u=u, accept_count=acc, proposal_count=prop)
        flags.append((acc - before).cpu().numpy())
        assert relerr(x[2].cpu().numpy(), G[k + kind + "_x"][i]) < 1e-9, i
    eng.check_status()
    flags = np.array(flags)
    for c in range(C):
        assert np.array_equal(flags[:, c], G[k + kind + "_accept"])  # INT path: bit-exact
    assert np.array_equal(prop.cpu().numpy(), np.full(C, zs.shape[0]))
    eng.close()


def test_mala_acceptance_and_stationarity():
    """cfg4-shaped target (d=500 correlated Gaussian, step 0.5): acceptance ~68 % and
    the chains stay in the target: E|L_Q'(x-mu)|^2 = d."""
    import torch

    d, C = 500, 256
    rng = np.random.default_rng(0)
    A = rng.standard_normal((d, 2 * d))
    Sig = A @ A.T / (2 * d)
    Qh = np.linalg.inv(Sig)
    Qh = (Qh + Qh.T) / 2
    eng = make_engine(C, seed=11)
    Q = eng.to_device(Qh)
    step = 0.5
    L, sl = eng.dense_cholesky(Q, 1.0 / step**2)
    x0 = np.linalg.solve(np.linalg.cholesky(Qh).T, rng.standard_normal((d, C))).T  # draws from the target
    x = eng.to_device(x0)
    acc = torch.zeros(C, dtype=torch.int64, device="cuda")
    prop = torch.zeros(C, dtype=torch.int64, device="cuda")
    n_steps = 60
    for it in range(n_steps):
        eng.mala_step(Q, None, L, sl, step, x, draw_index=it, accept_count=acc, proposal_count=prop)
    eng.check_status()
    rate = acc.sum().item() / prop.sum().item()
    assert 0.62 < rate < 0.74, rate
    xs = x.cpu().numpy()
    maha = np.einsum("ci,ij,cj->c", xs, Qh, xs)
    assert abs(maha.mean() / d - 1) < 0.02
    eng.close()
