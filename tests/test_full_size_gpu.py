"""BASELINE.json configs at their FULL sizes under pytest (size-independent properties; the path-wise parity of the same
routes is held at small sizes against the golden vectors): cfg2 Bayesian linear regression p = 1000, n = 10 000, 256
chains on both routes of the conjugate draw; cfg5 reversible jump + GMRF, 5000 nodes, n_max = 20, 512 chains."""

import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cfg2_full_size_both_routes_agree_and_solve_the_normal_equations():
    from openmcmc_amd.engine import Engine

    n, p, C = 10000, 1000, 256
    rng = np.random.default_rng(0)
    X = rng.standard_normal((n, p))
    beta = rng.standard_normal(p)
    y = X @ beta + 0.1 * rng.standard_normal(n)
    eng = Engine(C, seed=1)
    dX, dy = eng.to_device(X), eng.to_device(y)
    G, Xty = eng.gram(dX), eng.design_rhs(dX, dy)
    Gh = G.cpu().numpy()
    assert np.max(np.abs(Gh - X.T @ X)) < 1e-9 * np.max(np.abs(Gh))          # the MFMA Gram kernel at full size
    lam, tau = 0.005 + 0.01 * rng.random(C), 50 + 100 * rng.random(C)
    terms = [{"mat": None, "scale": eng.to_device(lam)}, {"mat": G, "rhs": Xty, "scale": eng.to_device(tau)}]
    V, ev = eng.dense_spectral_prepare(G)
    xs, ms, ls = eng.empty(C, p), eng.empty(C, p), eng.empty(C)
    eng.dense_spectral_sample(p, terms, 1, V, ev, xs, mean_out=ms, logdet_out=ls, draw_index=3)
    xc, mc, lc = eng.empty(C, p), eng.empty(C, p), eng.empty(C)
    eng.dense_sample_canonical(p, terms, xc, mean_out=mc, logdet_out=lc, draw_index=3)
    eng.check_status()
    ms_h, mc_h = ms.cpu().numpy(), mc.cpu().numpy()
    assert np.max(np.abs(ms_h - mc_h)) < 1e-9 * np.max(np.abs(mc_h))           # same conditional mean on both routes
    assert np.max(np.abs(ls.cpu().numpy() - lc.cpu().numpy()) / np.abs(lc.cpu().numpy())) < 1e-11
    XtY = X.T @ y
    for c in (0, 100, C - 1):                                                   # Q_c mu_c = b_c
        r = lam[c] * mc_h[c] + tau[c] * (Gh @ mc_h[c]) - tau[c] * XtY
        assert np.max(np.abs(r)) < 1e-8 * np.max(np.abs(tau[c] * XtY))
    # the draws scatter around the mean with the conditional covariance's scale (both routes, different square roots)
    for x, m in ((xs, ms), (xc, mc)):
        d = (x - m).cpu().numpy()
        q = lam[:, None] * d + tau[:, None] * (d @ Gh)
        maha = np.einsum("ci,ci->c", d, q)                                      # ~ chi^2_p per chain
        assert abs(maha.mean() - p) < 5 * np.sqrt(2 * p / C) + 1
    assert np.max(np.abs(ms_h.mean(0) - beta)) < 0.02                           # the posterior mean finds the coefficients
    eng.close()


def test_cfg5_full_size_runs_and_keeps_its_invariants():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from rj_problem import build, make_basis_host

    from openmcmc_amd import gmrf
    from openmcmc_amd.engine import Engine
    from openmcmc_amd.mcmc import MCMC

    n, n_max, C = 5000, 20, 512
    rng = np.random.default_rng(0)
    X = np.linspace(-10, 10, n)
    theta_true = np.array([[-6.0, -1.0, 4.5]])
    beta_true = np.array([[3.0], [-2.0], [4.0]])
    b_true = 0.05 * np.cumsum(rng.standard_normal(n)) * np.sqrt(48.0 / n)
    y = (make_basis_host(X.reshape(n, 1), theta_true) @ beta_true).ravel() + b_true + 0.1 * rng.standard_normal(n)
    P = gmrf.precision_irregular(np.arange(float(n))).tolil()
    P[0, 0] += 1e-3
    k0 = np.clip(rng.poisson(5, size=C), 1, n_max)
    eng = Engine(C, seed=1)
    mdl, state, samplers = build(y, X, P.tocsc(), n_max, eng, [rng.uniform(-10, 10, size=k) for k in k0],
                                 [rng.standard_normal(k) for k in k0], k0.astype(float))
    M = MCMC(state, samplers, model=mdl, n_burn=2, n_iter=4, n_chains=C, seed=1, engine=eng)
    M.run_mcmc()
    out = M.collect()
    nb = out["n_basis"][:, 0, :]
    assert np.all(nb >= 1) and np.all(nb <= n_max) and np.all(nb == np.round(nb))          # INT path
    assert np.all(np.isfinite(out["log_post"])) and np.all(out["tau"] > 0) and np.all(out["lambda"] > 0)
    theta = out["theta"]                                                                     # (C, n_max, n_iter), NaN beyond the live length
    for it in range(theta.shape[2]):
        live = np.arange(n_max)[None, :] < nb[:, it][:, None]
        assert np.all(np.isfinite(theta[:, :, it][live])) and np.all(np.isnan(theta[:, :, it][~live]))
        assert np.all(np.abs(theta[:, :, it][live]) <= 10.0)                                 # knots stay inside their limits
    assert np.any(nb[:, -1] != k0)                                                           # some chain has jumped
    assert 0 < samplers[4].accept_rate.accept.sum().item() < samplers[4].accept_rate.proposal.sum().item()
    eng.close()


def test_spectral_route_at_cfg2_size_has_the_factorisation_routes_conditional_law():
    """cfg2's production route is the spectral draw, which is not the path-wise image of the reference's (gmrf.py:481,462,434).
    At the config's full size (p = 1000, 256 chains with their own scales): (i) its mean and log det are the factorisation
    route's; (ii) per chain, the residual whitened with a square root of Q_c the TEST computes (numpy eigh of the Gram matrix,
    nothing of the route under test) is standard normal beyond its Mahalanobis mean: mean, variance, skewness, kurtosis, the
    KS distance of all coordinates over repeated draws, and no correlation between consecutive draws or neighbouring chains."""
    import torch
    from scipy import stats

    from openmcmc_amd.engine import Engine

    p, C, n_obs, draws = 1000, 256, 4000, 12
    rng = np.random.default_rng(11)
    X = rng.standard_normal((n_obs, p))
    y = X @ rng.standard_normal(p) + 0.5 * rng.standard_normal(n_obs)
    eng = Engine(C, seed=5)
    dX = eng.to_device(X)
    G, Xty = eng.gram(dX), eng.design_rhs(dX, eng.to_device(y))
    lam, tau = 0.05 + rng.random(C), 0.5 + 2 * rng.random(C)
    terms = [{"mat": None, "scale": eng.to_device(lam)}, {"mat": G, "rhs": Xty, "scale": eng.to_device(tau)}]
    V, ev = eng.dense_spectral_prepare(G)
    xs, ms, ls = eng.empty(C, p), eng.empty(C, p), eng.empty(C)
    xc, mc, lc = eng.empty(C, p), eng.empty(C, p), eng.empty(C)
    eng.dense_sample_canonical(p, terms, xc, mean_out=mc, logdet_out=lc, draw_index=1)
    eng.dense_spectral_sample(p, terms, 1, V, ev, xs, mean_out=ms, logdet_out=ls, draw_index=1)
    eng.check_status()
    assert (ms - mc).abs().max().item() < 1e-10 * mc.abs().max().item()
    assert ((ls - lc).abs() / lc.abs()).max().item() < 1e-12
    # the test's own square root: Q_c = U diag(lam_c + tau_c w) U'
    w_h, U_h = np.linalg.eigh(G.cpu().numpy())
    U = eng.to_device(U_h)
    root = torch.sqrt(eng.to_device(lam).reshape(C, 1) + eng.to_device(tau).reshape(C, 1) * eng.to_device(w_h).reshape(1, p))  # (C, p)
    W = []
    for k in range(draws):
        eng.dense_spectral_sample(p, terms, 1, V, ev, xs, mean_out=ms, draw_index=10 + k)
        W.append(((xs - ms) @ U) * root)  # rows: diag(sqrt(.)) U'(x - mu)
    eng.check_status()
    W = torch.stack(W).cpu().numpy()  # (draws, C, p)
    per_chain = W.transpose(1, 0, 2).reshape(C, draws * p)
    n = draws * p
    assert np.max(np.abs(per_chain.mean(axis=1))) < 5.0 / np.sqrt(n)
    assert np.max(np.abs(per_chain.var(axis=1) - 1.0)) < 5.0 * np.sqrt(2.0 / n)
    assert np.max(np.abs(stats.skew(per_chain, axis=1))) < 5.0 * np.sqrt(6.0 / n)
    assert np.max(np.abs(stats.kurtosis(per_chain, axis=1))) < 5.0 * np.sqrt(24.0 / n)
    assert stats.kstest(W[:, ::16].ravel(), "norm").pvalue > 1e-4
    # independence: consecutive draws of a chain, neighbouring chains of a draw, neighbouring coordinates
    m = W.size
    assert abs(np.mean(W[1:] * W[:-1])) < 5.0 / np.sqrt(m)
    assert abs(np.mean(W[:, 1:] * W[:, :-1])) < 5.0 / np.sqrt(m)
    assert abs(np.mean(W[:, :, 1:] * W[:, :, :-1])) < 5.0 / np.sqrt(m)
    eng.close()


@pytest.mark.parametrize("R,K,C", [(100, 100, 5), (100, 100, 520), (312, 32, 1100), (1250, 8, 1100), (666, 15, 2100)],
                         ids=["w100", "w100-many-chains", "w32-many-chains", "w8-many-chains", "w15-many-chains"])
def test_lattice_gmrf_at_full_size_solves_its_system_on_both_band_kernels(R, K, C):
    """SURVEY section 8f rank 1 at its size: a 100 x 100 lattice GMRF (10 000 nodes, bandwidth 100; gmrf.py:489-520 on a sparse
    precision of that shape), Q_c = lambda_c (L + kappa I) + tau_c I, a few chains.  Size-independent properties of the draw
    through the blocked kernel (omc_bandwide.hip): the mean solves Q_c mu_c = b_c, an injected z = 0 returns the mean itself, the
    draw minus the mean solves L' d = z (so Q d = L z has the right norm relation d' Q d = z' z), and mean, draw and log det
    agree with the column-at-a-time kernel.  The cases with more chains than the CUs hold workgroups at once run the forms of
    the kernel with several workgroups to a CU; it was the narrow one of them that showed a race in the backward pass (the
    solved block copied out of a ring that another wave was refilling): z = 0 no longer returned the mean."""
    from scipy import sparse

    from openmcmc_amd.engine import Engine

    # (more chains than CUs: the library takes the forms of the blocked kernel with several workgroups to a CU)
    n, w = R * K, K
    rng = np.random.default_rng(11)
    # 5-point Laplacian of the lattice in row-major order: bandwidth K
    ex, ey = np.ones(K), np.ones(R)
    Tx = sparse.diags([-ex[:-1], 2 * ex, -ex[:-1]], [-1, 0, 1])
    Ty = sparse.diags([-ey[:-1], 2 * ey, -ey[:-1]], [-1, 0, 1])
    Lap = (sparse.kron(sparse.identity(R), Tx) + sparse.kron(Ty, sparse.identity(K)) + 0.05 * sparse.identity(n)).tocsc()
    band = np.zeros((w + 1, n))
    for d in range(w + 1):
        band[d, : n - d] = Lap.diagonal(-d)
    lam, tau = 1.0 + rng.random(C), 0.5 + rng.random(C)
    y = rng.standard_normal(n)
    z = rng.standard_normal((C, n))
    eng = Engine(C, seed=4)
    terms = [{"band": eng.to_device(band), "scale": eng.to_device(lam)}, {"rhs": eng.to_device(y), "scale": eng.to_device(tau)}]
    out = {}
    for algo in (3, 2):
        eng.set_option("band_algo", algo)
        x, mu, ld, x0 = eng.empty(C, n), eng.empty(C, n), eng.empty(C), eng.empty(C, n)
        eng.band_sample_canonical(n, terms, x, z=eng.to_device(z), mean_out=mu, logdet_out=ld)
        eng.band_sample_canonical(n, terms, x0, z=eng.to_device(np.zeros((C, n))))
        eng.check_status()
        out[algo] = (x.cpu().numpy(), mu.cpu().numpy(), ld.cpu().numpy(), x0.cpu().numpy())
        if algo == 3:  # the same call again: the same bits (at this size and chain count an unfenced hand-over between waves shows)
            again = eng.empty(C, n)
            eng.band_sample_canonical(n, terms, again, z=eng.to_device(z))
            assert np.array_equal(again.cpu().numpy(), out[algo][0])
    xb, mb, lb, x0b = out[3]
    xo, mo, lo, _ = out[2]
    assert np.array_equal(x0b, mb)                                           # z = 0: the draw is the mean, bit for bit
    for c in sorted(set([0, 1, C // 2, C - 2, C - 1])):
        Q = lam[c] * Lap + tau[c] * sparse.identity(n)
        b = tau[c] * y
        assert np.max(np.abs(Q @ mb[c] - b)) < 1e-10 * np.max(np.abs(b))      # Q mu = b
        d = xb[c] - mb[c]
        assert abs(d @ (Q @ d) - z[c] @ z[c]) < 1e-10 * (z[c] @ z[c])         # d = L^-T z  =>  d' Q d = z' z
    scale = np.max(np.abs(xo))
    assert np.max(np.abs(xb - xo)) < 1e-11 * scale and np.max(np.abs(mb - mo)) < 1e-11 * scale
    assert np.max(np.abs(lb - lo)) < 1e-12 * np.max(np.abs(lo))
    eng.close()
