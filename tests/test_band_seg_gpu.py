"""The segmented lane-per-chain band kernel (k_band_lane_seg: up to 16 segments of a chain's columns, one wave each;
warm-up + checked joins for the pivots, exact affine maps for the two substitutions) against the same kernel in one piece
(band_algo = 1) and against a dense numpy factorisation: gmrf.sample_normal_canonical for banded precisions
(gmrf.py:167-198, 489-520) at sizes the one-piece kernel needs milliseconds for."""

import numpy as np
import pytest
from scipy import sparse

pytestmark = pytest.mark.gpu


def rw_band(n, order, ridge=1e-3):
    D = sparse.identity(n, format="csr")
    for _ in range(order):
        D = D[1:] - D[:-1]
    P = (D.T @ D + ridge * sparse.identity(n)).tocsc()
    band = np.zeros((order + 1, n))
    for d in range(order + 1):
        band[d, : n - d] = P.diagonal(-d)
    return P, band


def draw(n, C, order, lam, tau, algo, overlap=None, inject=True, seed=0, want_mean=True, ridge=1e-3):
    from openmcmc_amd.engine import Engine

    rng = np.random.default_rng(seed)
    eng = Engine(C, seed=4)
    eng.set_option("band_algo", algo)
    if overlap is not None:
        eng.set_option("band_seg_overlap", overlap)
    P, band = rw_band(n, order, ridge)
    t = np.arange(n) * 60.0 / n
    y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + rng.standard_normal(n)
    lam_c = lam * (0.5 + rng.random(C))
    tau_c = tau * (0.5 + rng.random(C))
    terms = [{"band": eng.to_device(band), "scale": eng.to_device(lam_c)}, {"rhs": eng.to_device(y), "scale": eng.to_device(tau_c)}]
    T = eng.band_terms(terms, n)
    z = rng.standard_normal((C, n)) if inject else None
    x, m, ld = eng.empty(C, n), (eng.empty(C, n) if want_mean else None), eng.empty(C)
    eng.band_sample_canonical(n, T, x, z=None if z is None else eng.to_device(z), draw_index=3, mean_out=m, logdet_out=ld)
    eng.check_status()
    fb = eng.counter("band_join_fallbacks")
    out = (x.cpu().numpy(), None if m is None else m.cpu().numpy(), ld.cpu().numpy(), fb, (P, y, lam_c, tau_c, z))
    eng.close()
    return out


@pytest.mark.parametrize("n,C,order,lam", [(10000, 70, 2, 100.0), (3000, 64, 2, 100.0), (4099, 5, 3, 100.0), (2500, 130, 1, 10.0)])
@pytest.mark.parametrize("inject", [True, False])
def test_segmented_band_draw_equals_the_one_piece_kernel(n, C, order, lam, inject):
    """(a first-order prior remembers longest: lam / tau = 10 keeps its pivots' memory inside the default warm-up)"""
    xs, ms, ls, fb, _ = draw(n, C, order, lam, 1.0, 0, inject=inject)
    x1, m1, l1, fb1, _ = draw(n, C, order, lam, 1.0, 1, inject=inject)
    assert fb == 0 and fb1 == 0  # the joins closed: this is the segmented route's result
    scale = np.abs(x1).max()
    assert np.abs(xs - x1).max() < 1e-10 * scale
    assert np.abs(ms - m1).max() < 1e-10 * scale
    assert np.abs(ls - l1).max() < 1e-9 * np.abs(l1).max()


def test_segmented_band_draw_against_dense_numpy():
    n, C, order = 2048, 3, 2
    x, m, ld, fb, (P, y, lam, tau, z) = draw(n, C, order, 50.0, 2.0, 0)
    assert fb == 0
    Pd = P.toarray()
    for c in range(C):
        Q = lam[c] * Pd + tau[c] * np.eye(n)
        L = np.linalg.cholesky(Q)
        u = np.linalg.solve(L, tau[c] * y)
        assert np.allclose(m[c], np.linalg.solve(L.T, u), rtol=0, atol=1e-9)
        assert np.allclose(x[c], np.linalg.solve(L.T, u + z[c]), rtol=0, atol=1e-9)
        assert abs(ld[c] - 2 * np.log(np.diag(L)).sum()) < 1e-8 * abs(ld[c])


def test_joins_that_do_not_close_are_retried_then_fall_back():
    """Long-memory priors: with lam / tau = 1e4 the pivots remember their start beyond the default warm-up but not beyond
    four times it -- the second attempt closes the joins; a likelihood 1e-9 times weaker than the prior (and next to no ridge) defeats
    that too and the group is factorised in one piece.  Either way the numbers are band_algo = 1's."""
    from openmcmc_amd.engine import Engine

    n, C, order = 6000, 64, 2
    for lam, tau, overlap, ridge, want_fallback in ((1e4, 1.0, 64, 1e-3, False), (1e5, 1e-4, 64, 1e-10, True)):
        xs, ms, ls, fb, _ = draw(n, C, order, lam, tau, 0, overlap=overlap, ridge=ridge)
        x1, m1, l1, _, _ = draw(n, C, order, lam, tau, 1, ridge=ridge)
        assert (fb >= 1) == want_fallback, (lam, tau, fb)
        scale = np.abs(x1).max()
        assert np.abs(xs - x1).max() < 1e-9 * scale and np.abs(ms - m1).max() < 1e-9 * scale
        assert np.abs(ls - l1).max() < 1e-9 * np.abs(l1).max()
    # the retry counter moves when the first attempt fails
    eng = Engine(C, seed=4)
    eng.set_option("band_seg_overlap", 64)
    P, band = rw_band(n, order)
    terms = [{"band": eng.to_device(band), "scale": eng.full((C,), 1e4)}, {"rhs": eng.to_device(np.ones(n)), "scale": eng.full((C,), 1.0)}]
    x = eng.empty(C, n)
    eng.band_sample_canonical(n, eng.band_terms(terms, n), x, draw_index=1)
    eng.check_status()
    assert eng.counter("band_join_retries") >= 1 and eng.counter("band_join_fallbacks") == 0
    eng.close()


def test_without_mean_output():
    xs, ms, ls, fb, _ = draw(5000, 66, 2, 100.0, 1.0, 0, want_mean=False)
    x1, _, l1, _, _ = draw(5000, 66, 2, 100.0, 1.0, 1, want_mean=False)
    assert ms is None and np.abs(xs - x1).max() < 1e-10 * np.abs(x1).max()


@pytest.mark.parametrize("n,C,order,algo", [(6000, 70, 2, 0), (4099, 5, 3, 0), (2500, 130, 1, 0), (700, 9, 2, 1), (300, 66, 5, 1)])
def test_per_chain_right_hand_side_on_the_lane_routes(n, C, order, algo):
    """A per-chain right-hand side (the offsets / sampled means of a hierarchical model on the band route) on the segmented
    route (algo 0 at these sizes) and on the one-piece lane kernel (algo 1, widths up to 8): transposed once so that a lane
    reads its value of a column coalesced, staged through registers and LDS a piece ahead in the segmented kernel.  Against
    the workgroup-per-chain kernel (band_algo = 2), which reads rhs_chain directly, and a dense solve for one chain."""
    from openmcmc_amd.engine import Engine

    rng = np.random.default_rng(n + C)
    P, band = rw_band(n, order, 1e-2)
    y = rng.standard_normal(n)
    lam_c, tau_c = 50.0 * (0.5 + rng.random(C)), 0.5 + rng.random(C)
    rc = rng.standard_normal((C, n + 3))[:, :n]  # a leading dimension that is not n
    z = rng.standard_normal((C, n))
    out = {}
    for a in (algo, 2):
        eng = Engine(C, seed=4)
        eng.set_option("band_algo", a)
        T = eng.band_terms([{"band": eng.to_device(band), "scale": eng.to_device(lam_c)},
                            {"rhs": eng.to_device(y), "scale": eng.to_device(tau_c)}], n)
        rcd = eng.to_device(np.pad(rc, ((0, 0), (0, 3))))[:, :n]
        x, m, ld = eng.empty(C, n), eng.empty(C, n), eng.empty(C)
        eng.band_sample_canonical(n, T, x, z=eng.to_device(z), rhs_chain=rcd, mean_out=m, logdet_out=ld)
        eng.check_status()
        assert eng.counter("band_join_fallbacks") == 0
        out[a] = (x.cpu().numpy(), m.cpu().numpy(), ld.cpu().numpy())
        eng.close()
    scale = np.abs(out[2][0]).max()
    assert np.abs(out[algo][0] - out[2][0]).max() < 1e-10 * scale
    assert np.abs(out[algo][1] - out[2][1]).max() < 1e-10 * scale
    assert np.abs(out[algo][2] - out[2][2]).max() < 1e-9 * np.abs(out[2][2]).max()
    c = C - 1
    Q = lam_c[c] * P.toarray() + tau_c[c] * np.eye(n)
    mu = np.linalg.solve(Q, tau_c[c] * y + rc[c])
    assert np.abs(out[algo][1][c] - mu).max() < 1e-8 * max(1.0, np.abs(mu).max())


def _engine_terms(n, C, order, lam=100.0, seed=0):
    from openmcmc_amd.engine import Engine

    rng = np.random.default_rng(seed)
    eng = Engine(C, seed=4)
    _, band = rw_band(n, order)
    t = np.arange(n) * 60.0 / n
    y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + rng.standard_normal(n)
    terms = [{"band": eng.to_device(band), "scale": eng.to_device(lam * (0.5 + rng.random(C)))},
             {"rhs": eng.to_device(y), "scale": eng.to_device(0.5 + rng.random(C))}]
    return eng, eng.band_terms(terms, n), rng


@pytest.mark.parametrize("n,C,order", [(5000, 70, 2), (3001, 129, 1)])
def test_rows_with_a_stride_keep_their_padding(n, C, order):
    """PHASE 2 writes the caller's rows itself (aligned tiles of 32 columns): rows longer than n, segment borders that cut
    a tile, a last group of fewer than 64 chains -- same numbers as into contiguous rows, nothing outside [0, n) touched."""
    eng, T, rng = _engine_terms(n, C, order)
    z = eng.to_device(rng.standard_normal((C, n)))
    x0, m0 = eng.empty(C, n), eng.empty(C, n)
    eng.band_sample_canonical(n, T, x0, z=z, mean_out=m0)
    pad = 37
    X, M = eng.full((C, n + pad), -7.0), eng.full((C, n + pad), -9.0)
    eng.band_sample_canonical(n, T, X[:, :n], z=z, mean_out=M[:, :n])
    eng.check_status()
    assert eng.counter("band_join_fallbacks") == 0
    assert np.array_equal(X[:, :n].cpu().numpy(), x0.cpu().numpy())
    assert np.array_equal(M[:, :n].cpu().numpy(), m0.cpu().numpy())
    assert (X[:, n:] == -7.0).all().item() and (M[:, n:] == -9.0).all().item()
    eng.close()


def test_the_draws_may_not_be_the_output_array():
    """(rows of x are written while other rows' draws are still being read: include/omcmc_hip.h)"""
    eng, T, rng = _engine_terms(600, 8, 2)
    x = eng.to_device(rng.standard_normal((8, 600)))
    with pytest.raises(ValueError):
        eng.band_sample_canonical(600, T, x, z=x)
    with pytest.raises(ValueError):
        eng.band_sample_canonical(600, T, x, mean_out=x)
    eng.close()


def test_full_size_draw_equals_the_one_piece_kernel():
    """RW2 at the headline's size (n = 10 000, 1024 chains: sixteen groups, one wave per SIMD) against the one-lane-per-chain
    kernel in one piece, generated draws."""
    xs, ms, ls, fb, _ = draw(10000, 1024, 2, 100.0, 1.0, 0, inject=False)
    x1, m1, l1, fb1, _ = draw(10000, 1024, 2, 100.0, 1.0, 1, inject=False)
    assert fb == 0 and fb1 == 0
    scale = np.abs(x1).max()
    assert np.abs(xs - x1).max() < 1e-10 * scale
    assert np.abs(ms - m1).max() < 1e-10 * scale
    assert np.abs(ls - l1).max() < 1e-9 * np.abs(l1).max()
