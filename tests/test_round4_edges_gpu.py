"""Edge sizes of the entry points added in round 4 (one chain, one node, one iteration, sizes around tile and block boundaries):
results against NumPy / the older kernels, and -- as important on this hardware -- no out-of-range access."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_engine(C, seed=0):
    from openmcmc_amd.engine import Engine

    return Engine(C, seed=seed)


def relerr(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("n_iter,C,size", [(1, 1, 1), (2, 1, 17), (5, 3, 8191), (3, 1, 8193), (2049, 1, 8200), (64, 130, 63), (4, 2, 16385)])
def test_store_summaries_at_edge_sizes(n_iter, C, size):
    rng = np.random.default_rng(n_iter + C + size)
    a = rng.standard_normal((n_iter, C, size))
    eng = make_engine(C)
    t = eng.to_device(a)
    q = [0.0, 0.3, 1.0]
    assert np.array_equal(eng.store_quantiles(t, q, pooled=False).cpu().numpy(), np.quantile(a, q, axis=0))
    assert np.array_equal(eng.store_quantiles(t, q, pooled=True).cpu().numpy(), np.quantile(a.reshape(-1, size), q, axis=0))
    mean, var = eng.store_moments(t, pooled=True)
    flat = a.reshape(-1, size)
    assert relerr(mean.cpu().numpy(), flat.mean(axis=0)) < 1e-12
    if flat.shape[0] > 1:
        assert relerr(var.cpu().numpy(), flat.var(axis=0, ddof=1)) < 1e-10
    for every, first in ((1, 0), (2, 0), (3, min(1, n_iter - 1)), (n_iter, n_iter - 1)):
        assert np.array_equal(eng.store_thin(t, every, first=first).cpu().numpy(), a[first::every])
    eng.check_status()
    eng.close()


@pytest.mark.parametrize("d,C,steps", [(1, 1, 1), (2, 3, 33), (511, 2, 32), (512, 1, 31), (513, 2, 3), (1024, 1, 2), (2047, 1, 2), (2048, 2, 1)])
def test_mala_run_white_at_edge_sizes(d, C, steps):
    import torch

    rng = np.random.default_rng(d + C)
    A = rng.standard_normal((d, d + 3))
    Qh = A @ A.T / (d + 3) + 0.5 * np.eye(d)
    step = 0.4
    x0 = rng.standard_normal((C, d)) * 0.5
    eng = make_engine(C, seed=2)
    L, sl = eng.dense_cholesky(eng.to_device(Qh), 1.0 / step**2)
    xa, xb = eng.to_device(x0), eng.to_device(x0)
    acc_a, acc_b = (torch.zeros(C, dtype=torch.int64, device="cuda") for _ in range(2))
    for i in range(steps):
        eng.mala_step_white(None, L, sl, step, xa, state_is_current=i > 0, draw_index=7 + i, accept_count=acc_a)
    xs, lps = eng.empty(steps, C, d), eng.empty(steps, C)
    eng.mala_run_white(None, L, sl, step, xb, steps, draw_index0=7, draw_stride=1, x_store=xs, logp_store=lps, accept_count=acc_b)
    eng.check_status()
    assert np.array_equal(acc_a.cpu().numpy(), acc_b.cpu().numpy())
    assert relerr(xb.cpu().numpy(), xa.cpu().numpy()) < 1e-11
    assert np.array_equal(xs[-1].cpu().numpy(), xb.cpu().numpy())
    eng.mala_run_white(None, L, sl, step, xb, 0)  # zero steps: nothing happens
    eng.check_status()
    assert np.array_equal(xs[-1].cpu().numpy(), xb.cpu().numpy())
    eng.close()


@pytest.mark.parametrize("n,w,C", [(1, 1, 1), (2, 1, 3), (15, 14, 2), (16, 15, 1), (17, 16, 65), (31, 9, 1), (33, 32, 2), (129, 128, 1), (48, 3, 70)])
def test_blocked_band_kernel_at_edge_sizes(n, w, C):
    """(lengths below, at and just above a block; bandwidth n - 1; chain counts that are not a multiple of anything)"""
    from scipy import sparse

    rng = np.random.default_rng(1000 * n + w)
    B = sparse.diags([rng.standard_normal(n - d) * 0.3 for d in range(w + 1)], list(range(0, -w - 1, -1)), shape=(n, n)).toarray()
    M = B @ B.T + np.eye(n)
    wM = min(n - 1, 2 * w) if n > 1 else 0
    wM = min(wM, 128)
    band = np.zeros((wM + 1, n))
    for d in range(wM + 1):
        band[d, : n - d] = np.diagonal(M, -d)
    Mb = sum(np.diag(band[d, : n - d], -d) + (np.diag(band[d, : n - d], d) if d else 0) for d in range(wM + 1))
    eng = make_engine(C)
    z = rng.standard_normal((C, n))
    extra = rng.standard_normal((C, n))
    terms = [{"band": eng.to_device(band)}]
    out = {}
    for algo in (3, 2):
        eng.set_option("band_algo", algo)
        x, mu, ld = eng.empty(C, n), eng.empty(C, n), eng.empty(C)
        eng.band_sample_canonical(n, terms, x, z=eng.to_device(z), rhs_chain=eng.to_device(extra), mean_out=mu, logdet_out=ld)
        eng.check_status()
        out[algo] = (x.cpu().numpy(), mu.cpu().numpy(), ld.cpu().numpy())
    L = np.linalg.cholesky(Mb)
    for c in range(C):
        mu = np.linalg.solve(Mb, extra[c])
        xo = mu + np.linalg.solve(L.T, z[c])
        assert relerr(out[3][0][c], xo) < 1e-9 and relerr(out[3][1][c], mu) < 1e-9
        assert abs(out[3][2][c] - 2 * np.sum(np.log(np.diag(L)))) < 1e-10 * max(1.0, abs(out[3][2][c]))
    assert relerr(out[3][0], out[2][0]) < 1e-10
    eng.close()


@pytest.mark.parametrize("n,C", [(1, 1), (2, 65), (31, 3), (32, 1), (33, 64), (63, 2), (64, 1), (65, 130), (97, 5)])
def test_truncated_scan_at_edge_sizes(n, C):
    """The four-wave scan on lengths around its rounds of 32 sites and blocks of 64, chain counts around the 64 lanes of a group:
    against a straight NumPy restatement of gmrf.py:201-266 with the same uniforms."""
    from scipy import stats

    rng = np.random.default_rng(n * 7 + C)
    d = np.full(n, 2.0) + rng.random(n)
    off = -rng.random(max(n - 1, 0)) * 0.9
    y = rng.standard_normal(n)
    lam, tau = 1.0 + rng.random(C), 0.5 + rng.random(C)
    lo, hi = -0.5 * np.ones(n), 1.5 * np.ones(n)
    u = rng.random((C, n)) * 0.98 + 0.01
    x0 = rng.random((C, n))
    eng = make_engine(C)
    terms = [{"diag": eng.to_device(d), "off": eng.to_device(off) if n > 1 else None, "scale": eng.to_device(lam)},
             {"rhs": eng.to_device(y), "scale": eng.to_device(tau)}]
    T = eng.tridiag_terms(terms, n)
    x = eng.to_device(x0)
    eng.tridiag_gibbs_truncated(n, T, x, lower=eng.to_device(lo), upper=eng.to_device(hi), u=eng.to_device(u))
    eng.check_status()
    got = x.cpu().numpy()
    for c in range(C):
        xr = x0[c].copy()
        for i in range(n):
            a = lam[c] * d[i] + tau[c]
            b = tau[c] * y[i]
            row = a * xr[i]
            if i > 0:
                row += lam[c] * off[i - 1] * xr[i - 1]
            if i + 1 < n:
                row += lam[c] * off[i] * xr[i + 1]
            mean = b / a if n == 1 else (b - row + a * xr[i]) / a
            sd = 1 / np.sqrt(a)
            al, be = (lo[i] - mean) / sd, (hi[i] - mean) / sd
            xr[i] = stats.truncnorm.ppf(u[c, i], al, be) * sd + mean
        assert relerr(got[c], xr) < 1e-9, c
    eng.close()


@pytest.mark.parametrize("n,w", [(1, 1), (2, 1), (3, 2), (5, 4), (5, 9), (8, 12), (17, 16), (33, 40), (16, 15), (32, 31), (48, 5)])
@pytest.mark.parametrize("algo", [0, 3])
def test_band_draw_on_chains_shorter_than_the_band(n, w, algo):
    """Declared bandwidths up to and beyond n - 1 (a dense matrix in band storage), lengths of one block and one more, on the
    automatic choice of kernel and on the blocked one: draw and log det against a dense Cholesky factor."""
    from openmcmc_amd.engine import Engine

    rng = np.random.default_rng(1000 * n + w)
    A = rng.standard_normal((n, n))
    M = A @ A.T
    M[np.abs(np.subtract.outer(np.arange(n), np.arange(n))) > w] = 0.0
    M += (np.abs(M).sum(1).max() + 1.0) * np.eye(n)
    band = np.zeros((w + 1, n))
    for d in range(min(w, n - 1) + 1):
        band[d, : n - d] = np.diag(M, -d)
    C = 3
    eng = Engine(C, seed=1)
    eng.set_option("band_algo", algo)
    z, b = rng.standard_normal((C, n)), rng.standard_normal(n)
    x, ld = eng.empty(C, n), eng.empty(C)
    eng.band_sample_canonical(n, [{"band": eng.to_device(band), "rhs": eng.to_device(b)}], x, z=eng.to_device(z), logdet_out=ld)
    eng.check_status()
    L = np.linalg.cholesky(M)
    want = np.linalg.solve(L.T, np.linalg.solve(L, b)[:, None] + z.T).T
    assert np.max(np.abs(x.cpu().numpy() - want)) < 1e-12 * max(1.0, np.abs(want).max())
    assert np.max(np.abs(ld.cpu().numpy() - 2 * np.log(np.diag(L)).sum())) < 1e-11 * max(1.0, abs(2 * np.log(np.diag(L)).sum()))
    eng.close()
