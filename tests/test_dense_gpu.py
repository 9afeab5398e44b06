"""Dense Normal-Normal path (regression-shaped conditionals, BASELINE configs[0]/[1] shape) through
the C ABI against the reference's golden vectors and the oracle."""

import numpy as np
import pytest

from oracle import gmrf_ref, sweep_ref

pytestmark = pytest.mark.gpu
TOL = 1e-10


def relerr(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def make_engine(C, **kw):
    from openmcmc_amd.engine import Engine

    return Engine(C, **kw)


@pytest.mark.parametrize("p", [1, 7, 32])
def test_dense_primitives_golden(golden, p):
    """gmrf.cholesky / cho_solve / sample_normal_canonical on a dense Q (reference vectors)."""
    G = golden("dense_primitives")
    k = f"p{p}_"
    C = 3
    eng = make_engine(C)
    Q = eng.to_device(G[k + "Q"])
    terms = [{"mat": Q, "scale": eng.full((C,), 1.0)}]
    b = eng.to_device(np.tile(G[k + "b"], (C, 1)))
    z = eng.to_device(np.tile(G[k + "z"], (C, 1)))
    x, mean, logdet = eng.empty(C, p), eng.empty(C, p), eng.empty(C)
    eng.dense_sample_canonical(p, terms, x, z=z, rhs_chain=b, mean_out=mean, logdet_out=logdet)
    eng.check_status()
    for c in range(C):
        assert relerr(x[c].cpu().numpy(), G[k + "x"]) < TOL
        assert relerr(mean[c].cpu().numpy(), G[k + "mu"]) < TOL
        assert relerr(logdet[c].item(), 2 * np.sum(np.log(np.diag(G[k + "L"])))) < TOL
    eng.close()


@pytest.mark.parametrize("tag", ["ex3", "p7"])
def test_linreg_chain_golden(golden, tag):
    """MCMC.run_mcmc of example 3 (BASELINE configs[0]) replayed through the ABI with the reference's
    draws: beta, tau, lambda, log_post and the fitted values y = X beta."""
    G = golden("linreg_chain")
    k = tag + "_"
    X, y = G[k + "X"], G[k + "y"]
    N, p = X.shape
    n_burn, n_iter = int(G[k + "n_burn"]), int(G[k + "n_iter"])
    C = 2
    eng = make_engine(C)
    dX, dy = eng.to_device(X), eng.to_device(y)
    Gram, Xty = eng.gram(dX), eng.design_rhs(dX, dy)
    lam, tau = eng.full((C,), 0.01), eng.full((C,), 1.0)
    terms = eng.dense_terms([{"mat": None, "scale": lam}, {"mat": Gram, "rhs": Xty, "scale": tau}], p)
    ident = eng.tridiag_terms([{}], p)  # (beta - 0)' I (beta - 0)
    beta, fitted = eng.empty(C, p), eng.empty(C, N)
    q_tau, q_lam, lp = eng.empty(C), eng.empty(1, C), eng.empty(C)
    zero = eng.zeros(1)
    store = {key: [] for key in ("beta", "tau", "lambda", "log_post", "y")}
    for it in range(n_burn + n_iter):
        eng.dense_sample_canonical(p, terms, beta, z=eng.to_device(np.tile(G[k + "z"][it], (C, 1))))
        eng.design_predict(dX, beta, fitted)
        eng.weighted_resid_sq(dy, fitted, q_tau)
        g = G[k + "g"][it]
        eng.normal_gamma_update(1e-3, 1e-3, N, q_tau, tau, g=eng.full((C,), g[0]))
        eng.tridiag_quadform(p, ident, beta, q_lam)
        eng.normal_gamma_update(1e-3, 1e-3, p, q_lam[0], lam, g=eng.full((C,), g[1]))
        if it < n_burn:
            continue
        eng.scaled_gauss_logpdf(N, tau, zero, q_tau, lp)
        eng.scaled_gauss_logpdf(p, lam, zero, q_lam[0], lp, accumulate=True)
        eng.gamma_logpdf(tau, 1e-3, 1e-3, lp, accumulate=True)
        eng.gamma_logpdf(lam, 1e-3, 1e-3, lp, accumulate=True)
        store["beta"].append(beta[1].cpu().numpy().copy())
        store["tau"].append(tau[1].item())
        store["lambda"].append(lam[1].item())
        store["log_post"].append(lp[1].item())
        store["y"].append(fitted[1].cpu().numpy().copy())
    eng.check_status()
    assert relerr(np.array(store["beta"]).T, G[k + "store_beta"]) < 1e-9
    assert relerr(store["tau"], G[k + "store_tau"].ravel()) < 1e-9
    assert relerr(store["lambda"], G[k + "store_lambda"].ravel()) < 1e-9
    assert relerr(store["log_post"], G[k + "store_log_post"].ravel()) < 1e-9
    assert relerr(np.array(store["y"]).T, G[k + "store_y"]) < 1e-9
    eng.close()


def test_dense_random_vs_oracle_and_helpers():
    """Mid-size regression block (n=600, p=150, 5 chains): Gram, X'Wy, draw, mean, log det, fitted
    values and weighted residuals against the oracle."""
    rng = np.random.default_rng(3)
    n, p, C = 600, 150, 5
    X = rng.standard_normal((n, p))
    w = 0.5 + rng.random(n)
    y = X @ rng.standard_normal(p) + 0.1 * rng.standard_normal(n)
    Pm = np.diag(0.5 + rng.random(p))
    mu = rng.standard_normal(p)
    lam, tau = 0.5 + rng.random(C), 1 + rng.random(C)
    z = rng.standard_normal((C, p))
    eng = make_engine(C)
    dX, dy, dw = eng.to_device(X), eng.to_device(y), eng.to_device(w)
    Gram, Xtwy = eng.gram(dX, dw), eng.design_rhs(dX, dy, dw)
    assert relerr(Gram.cpu().numpy(), X.T @ (w[:, None] * X)) < 1e-12
    assert relerr(Xtwy.cpu().numpy(), X.T @ (w * y)) < 1e-12
    terms = [{"mat": eng.to_device(Pm), "rhs": eng.to_device(Pm @ mu), "scale": eng.to_device(lam)},
             {"mat": Gram, "rhs": Xtwy, "scale": eng.to_device(tau)}]
    x, mean, logdet = eng.empty(C, p), eng.empty(C, p), eng.empty(C)
    eng.dense_sample_canonical(p, terms, x, z=eng.to_device(z), mean_out=mean, logdet_out=logdet)
    fitted = eng.design_predict(dX, x)
    rs = eng.empty(C)
    eng.weighted_resid_sq(dy, fitted, rs, w=dw)
    eng.check_status()
    for c in range(C):
        Q = lam[c] * Pm + tau[c] * (X.T @ (w[:, None] * X))
        b = lam[c] * (Pm @ mu) + tau[c] * (X.T @ (w * y))
        xo, mo, L = gmrf_ref.draw_canonical(b.reshape(p, 1), Q, z[c])
        assert relerr(x[c].cpu().numpy(), xo.ravel()) < 1e-9
        assert relerr(mean[c].cpu().numpy(), mo.ravel()) < 1e-9
        assert relerr(logdet[c].item(), 2 * np.sum(np.log(np.diag(L)))) < TOL
        f = X @ x[c].cpu().numpy()
        assert relerr(fitted[c].cpu().numpy(), f) < 1e-12
        assert relerr(rs[c].item(), np.sum(w * (y - f) ** 2)) < 1e-11
    # in-kernel draws equal fill_normal's stream
    x2 = eng.empty(C, p)
    zz = eng.fill_normal(p, draw_index=9)
    eng.dense_sample_canonical(p, terms, x, z=zz)
    eng.dense_sample_canonical(p, terms, x2, z=None, draw_index=9)
    assert relerr(x2.cpu().numpy(), x.cpu().numpy()) < 1e-12
    eng.close()


def test_dense_not_positive_definite():
    eng = make_engine(3)
    p = 20
    A = np.eye(p)
    A[3, 3] = -1.0
    terms = [{"mat": eng.to_device(A), "scale": eng.to_device(np.array([1.0, 1.0, 1.0]))},
             {"mat": None, "scale": eng.to_device(np.array([2.0, 0.5, 2.0]))}]  # chain 1: 0.5 - 1 < 0
    x = eng.empty(3, p)
    eng.dense_sample_canonical(p, terms, x, z=eng.zeros(3, p))
    with pytest.raises(np.linalg.LinAlgError, match="chain 1"):
        eng.check_status()
    eng.close()


@pytest.mark.parametrize("p", [256, 300, 1000])
def test_blocked_cholesky_route(p):
    """Orders >= 256 take the own left-looking blocked Cholesky (batched DGEMM updates + k_chol_panel); it must give
    the oracle's draw, mean and log-determinant (numpy LAPACK, natural order), agree with the rocSOLVER route it
    replaces, and latch a non-positive-definite chain."""
    from oracle import gmrf_ref

    rng = np.random.default_rng(p)
    C, n = 3, 2 * p
    X = rng.standard_normal((n, p))
    G = X.T @ X
    y = rng.standard_normal(n)
    lam, tau = rng.random(C) + 0.5, rng.random(C) * 2 + 0.5
    z = rng.standard_normal((C, p))
    eng = make_engine(C)
    terms = eng.dense_terms([{"mat": None, "scale": eng.to_device(lam)},
                             {"mat": eng.to_device(G), "rhs": eng.to_device(X.T @ y), "scale": eng.to_device(tau)}], p)
    x, mean, logdet = eng.empty(C, p), eng.empty(C, p), eng.empty(C)
    eng.dense_sample_canonical(p, terms, x, z=eng.to_device(z), mean_out=mean, logdet_out=logdet)
    eng.check_status()
    eng.set_option("dense_use_rocsolver", 1)
    x2, mean2, logdet2 = eng.empty(C, p), eng.empty(C, p), eng.empty(C)
    eng.dense_sample_canonical(p, terms, x2, z=eng.to_device(z), mean_out=mean2, logdet_out=logdet2)
    eng.check_status()
    eng.set_option("dense_use_rocsolver", 0)
    xg, mg, lg = x.cpu().numpy(), mean.cpu().numpy(), logdet.cpu().numpy()
    for c in range(C):
        Q = lam[c] * np.eye(p) + tau[c] * G
        xo, mo, L = gmrf_ref.draw_canonical((tau[c] * (X.T @ y)).reshape(p, 1), Q, z[c].reshape(p, 1))
        scale = max(1.0, np.abs(xo).max())
        assert np.max(np.abs(xg[c] - xo.ravel())) / scale < 1e-10
        assert np.max(np.abs(mg[c] - mo.ravel())) / scale < 1e-10
        assert abs(lg[c] - 2 * np.sum(np.log(np.diag(L)))) < 1e-10 * abs(lg[c])
    assert np.max(np.abs(xg - x2.cpu().numpy())) < 1e-10 * max(1.0, np.abs(xg).max())
    assert np.max(np.abs(lg - logdet2.cpu().numpy())) < 1e-10 * np.abs(lg).max()
    bad = eng.dense_terms([{"mat": eng.to_device(G), "scale": eng.to_device(np.array([1.0, -1.0, 1.0]))}], p)
    eng.dense_sample_canonical(p, bad, x, z=eng.to_device(z))
    with pytest.raises(np.linalg.LinAlgError, match="chain 1"):
        eng.check_status()
    eng.close()


@pytest.mark.parametrize("C", [64, 129])
def test_blocked_cholesky_two_halves_on_two_streams(C):
    """From 64 chains on, the blocked factorisation runs the two halves of the chains on two streams (one half's panel
    kernels under the other half's update GEMMs, fork / join by events).  Same arithmetic per chain: draws, means and log
    determinants are bit-identical to the one-batch run; a chain that is not positive definite in the SECOND half is reported
    under its own index; and the caller's stream sees the result complete (the next launch on it reads x)."""
    import torch

    p = 320
    rng = np.random.default_rng(C)
    X = rng.standard_normal((2 * p, p))
    G = X.T @ X / p
    lam, tau = rng.random(C) + 0.5, rng.random(C) * 2 + 0.5
    eng = make_engine(C, seed=4)
    terms = eng.dense_terms([{"mat": None, "scale": eng.to_device(lam)},
                             {"mat": eng.to_device(G), "rhs": eng.to_device(rng.standard_normal(p)), "scale": eng.to_device(tau)}], p)
    out = {}
    for overlap in (0, 1):
        eng.set_option("dense_overlap", overlap)
        x, mean, logdet = eng.empty(C, p), eng.empty(C, p), eng.empty(C)
        eng.dense_sample_canonical(p, terms, x, draw_index=7, mean_out=mean, logdet_out=logdet)
        total = x.sum()  # a torch kernel on the same stream right behind the library call
        eng.check_status()
        out[overlap] = (x.clone(), mean.clone(), logdet.clone(), total.item())
    for a, b in zip(out[0][:3], out[1][:3]):
        assert torch.equal(a, b)
    assert out[0][3] == out[1][3]
    # oracle on a chain of each half
    for c in (0, C - 1):
        Q = lam[c] * np.eye(p) + tau[c] * G
        assert abs(out[1][2][c].item() - np.linalg.slogdet(Q)[1]) < 1e-10 * abs(np.linalg.slogdet(Q)[1])
    sc = np.ones(C)
    sc[C - 2] = -1.0
    bad = eng.dense_terms([{"mat": eng.to_device(G), "scale": eng.to_device(sc)}], p)
    eng.dense_sample_canonical(p, bad, eng.empty(C, p))
    with pytest.raises(np.linalg.LinAlgError, match=f"chain {C - 2}"):
        eng.check_status()
    eng.close()


# ---- spectral route: Q_c = a_c I + b_c M in M's eigenbasis (omc_dense_spectral_sample) ------------------------------
@pytest.mark.parametrize("p,C", [(70, 5), (300, 4)])
def test_spectral_route_mean_logdet_and_mahalanobis(p, C):
    """Same mean and log det as the factorisation route (the reference's values, gmrf.py:196, 339) and, for injected
    draws, (x - mu)' Q (x - mu) = |z|^2 exactly: x - mu = V D^-1/2 z is a square-root image of z like L^-T z is."""
    from openmcmc_amd.engine import Engine

    rng = np.random.default_rng(p)
    n_obs = 3 * p
    X = rng.standard_normal((n_obs, p))
    y = rng.standard_normal(n_obs)
    eng = Engine(C, seed=4)
    dX = eng.to_device(X)
    G, Xty = eng.gram(dX), eng.design_rhs(dX, eng.to_device(y))
    lam, tau = 0.3 + rng.random(C), 0.5 + 2 * rng.random(C)
    terms = [{"mat": None, "scale": eng.to_device(lam)}, {"mat": G, "rhs": Xty, "scale": eng.to_device(tau)}]
    extra = rng.standard_normal((C, p))
    z = rng.standard_normal((C, p))
    V, ev = eng.dense_spectral_prepare(G)
    xs, ms, ls = eng.empty(C, p), eng.empty(C, p), eng.empty(C)
    eng.dense_spectral_sample(p, terms, 1, V, ev, xs, z=eng.to_device(z), rhs_chain=eng.to_device(extra), mean_out=ms, logdet_out=ls)
    xc, mc, lc = eng.empty(C, p), eng.empty(C, p), eng.empty(C)
    eng.dense_sample_canonical(p, terms, xc, z=eng.to_device(z), rhs_chain=eng.to_device(extra), mean_out=mc, logdet_out=lc)
    eng.check_status()
    assert np.max(np.abs(ms.cpu().numpy() - mc.cpu().numpy())) / np.max(np.abs(mc.cpu().numpy())) < 1e-11
    assert np.max(np.abs(ls.cpu().numpy() - lc.cpu().numpy()) / np.abs(lc.cpu().numpy())) < 1e-12
    Gh = G.cpu().numpy()
    for c in range(C):
        Q = lam[c] * np.eye(p) + tau[c] * Gh
        r = xs[c].cpu().numpy() - ms[c].cpu().numpy()
        assert abs(r @ Q @ r - z[c] @ z[c]) < 1e-10 * (z[c] @ z[c])
        mu = np.linalg.solve(Q, tau[c] * (X.T @ y) + extra[c])
        assert np.max(np.abs(ms[c].cpu().numpy() - mu)) < 1e-10 * np.max(np.abs(mu))
    eng.close()


def test_spectral_route_draws_have_the_right_law():
    """In-kernel draws: covariance of x - mu over many chains against Q^-1, chi^2 law of the Mahalanobis distance."""
    from scipy import stats

    from openmcmc_amd.engine import Engine

    p, C = 64, 20000
    rng = np.random.default_rng(2)
    X = rng.standard_normal((200, p))
    eng = Engine(C, seed=9)
    G = eng.gram(eng.to_device(X))
    lam, tau = 0.7, 1.3
    terms = [{"mat": None, "scale": eng.full((C,), lam)}, {"mat": G, "rhs": eng.to_device(rng.standard_normal(p)), "scale": eng.full((C,), tau)}]
    V, ev = eng.dense_spectral_prepare(G)
    x, m = eng.empty(C, p), eng.empty(C, p)
    eng.dense_spectral_sample(p, terms, 1, V, ev, x, mean_out=m, draw_index=3)
    eng.check_status()
    r = (x - m).cpu().numpy()
    Q = lam * np.eye(p) + tau * G.cpu().numpy()
    maha = np.einsum("ci,ij,cj->c", r, Q, r)
    assert stats.kstest(maha, "chi2", args=(p,)).pvalue > 1e-3
    cov, S = r.T @ r / C, np.linalg.inv(Q)
    assert np.max(np.abs(cov - S)) < 6 * np.max(np.abs(S)) / np.sqrt(C)
    eng.close()


@pytest.mark.parametrize("n,p,C", [(1000, 10000, 7), (200, 4096, 256), (37, 5003, 3), (300, 4095, 5)])
def test_design_predict_long_contraction(n, p, C):
    """omc_design_predict with a short output under a long contraction (A'W applied to per-chain vectors of the observation
    space: the per-chain offsets of the dense route) cuts the contraction into slices and adds them in order; against numpy,
    including a slice count that does not divide the contraction and a size just below the switch."""
    rng = np.random.default_rng(n + p)
    X, B = rng.standard_normal((n, p)), rng.standard_normal((C, p))
    eng = make_engine(C)
    got = eng.design_predict(eng.to_device(X), eng.to_device(B)).cpu().numpy()
    ref = B @ X.T
    assert np.max(np.abs(got - ref)) < 1e-11 * np.abs(ref).max()
    again = eng.design_predict(eng.to_device(X), eng.to_device(B)).cpu().numpy()
    assert np.array_equal(got, again)
    eng.close()
