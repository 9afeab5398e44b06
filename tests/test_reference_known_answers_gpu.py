"""The reference's own closed-form sampler tests (tests/test_sampler.py:229-341 of openMCMC: draws
mocked to zeros / ones / the Gamma mean) and the MCMC call-count test (tests/test_mcmc.py:83-124),
re-expressed on the chain-batched API with the `inject` hooks."""

import numpy as np
import pytest
from scipy import sparse

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def make_engine(C, seed=0):
    from openmcmc_amd.engine import Engine

    return Engine(C, seed=seed)


def regression_setup(n_resp, n_par, C, seed=0):
    """The reference's shared test model, Normal-Normal part: response ~ N(X parameter, (tau P)^-1),
    parameter ~ N(prior_mean, (lambda I)^-1)."""
    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.distribution import Gamma
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.engine import Engine
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix

    rng = np.random.default_rng(seed)
    mdl = Model([
        Normal("response", mean=LinearCombination(form={"parameter": "prefactor_matrix"}),
               precision=ScaledMatrix(matrix="response_precision", scalar="tau")),
        Normal("parameter", mean="prior_mean", precision=ScaledMatrix(matrix="prior_precision", scalar="lambda")),
        Gamma("tau", shape="gamma_shape", rate="gamma_rate"),
    ])
    eng = Engine(C)
    state = {
        "response": rng.standard_normal((n_resp, 1)),
        "prefactor_matrix": rng.standard_normal((n_resp, n_par)),
        "response_precision": sparse.diags(0.5 + rng.random(n_resp), format="csc"),
        "tau": 1.7,
        "prior_mean": rng.standard_normal((n_par, 1)),
        "prior_precision": sparse.identity(n_par, format="csc"),
        "lambda": 0.8,
        "parameter": ChainArray(eng.to_device(rng.standard_normal((C, n_par)))),
        "gamma_shape": 2.0, "gamma_rate": 3.0,
    }
    from openmcmc_amd.chains import host_2d

    for k, v in list(state.items()):
        if not sparse.issparse(v) and not isinstance(v, ChainArray):
            state[k] = host_2d(v)
    return mdl, eng, state


@pytest.mark.parametrize("n_resp,n_par", [(20, 3), (1, 1), (50, 10)])
def test_normalnormal_closed_forms(n_resp, n_par):
    """check_normalnormal (tests/test_sampler.py:262-308)."""
    from openmcmc_amd.sampler.sampler import NormalNormal

    C = 2
    mdl, eng, state = regression_setup(n_resp, n_par, C)
    zeros = lambda s_, t: eng.zeros(C, n_par)  # noqa: E731
    ones = lambda s_, t: eng.full((C, n_par), 1.0)  # noqa: E731

    # 1) all-zero design, no randomness -> the prior mean
    st = dict(state)
    st["prefactor_matrix"] = np.zeros_like(state["prefactor_matrix"])
    smp = NormalNormal("parameter", mdl).bind(eng)
    smp.inject = zeros
    out = smp.sample(st)
    for c in range(C):
        assert np.allclose(out["parameter"].chain(c), state["prior_mean"])

    # 2) zero prior precision -> the (generalised) least-squares solution
    if n_resp > 1:
        st = dict(state)
        st["lambda"] = np.array([[0.0]])
        smp = NormalNormal("parameter", mdl).bind(eng)
        smp.inject = zeros
        out = smp.sample(st)
        X, W = state["prefactor_matrix"], 1.7 * state["response_precision"].toarray()
        comparison = np.linalg.solve(X.T @ W @ X, X.T @ W @ state["response"])
        assert np.allclose(out["parameter"].chain(1), comparison)

    # 3) zero means, draws all ones -> x = (chol(X'QX + P)')^-1 1
    st = dict(state)
    st["response"] = np.zeros_like(state["response"])
    st["prior_mean"] = np.zeros_like(state["prior_mean"])
    smp = NormalNormal("parameter", mdl).bind(eng)
    smp.inject = ones
    out = smp.sample(st)
    X, W = state["prefactor_matrix"], 1.7 * state["response_precision"].toarray()
    comparison = np.linalg.solve(np.linalg.cholesky(X.T @ W @ X + 0.8 * np.eye(n_par)).T, np.ones((n_par, 1)))
    assert np.allclose(out["parameter"].chain(0), comparison)
    eng.check_status()
    eng.close()


def test_normalgamma_recovers_mean_squared_residual():
    """check_normalgamma (tests/test_sampler.py:311-341): gamma draw mocked to its mean a*scale, prior
    shape and rate zero => 1/tau = mean(r' P r / n)."""
    from openmcmc_amd.sampler.sampler import NormalGamma

    C, n_resp, n_par = 3, 40, 4
    mdl, eng, state = regression_setup(n_resp, n_par, C, seed=5)
    state["gamma_shape"], state["gamma_rate"] = np.array([[0.0]]), np.array([[0.0]])
    state["response_precision"] = sparse.identity(n_resp, format="csc")
    smp = NormalGamma("tau", mdl).bind(eng)
    a_post = n_resp / 2
    smp.inject = lambda s_, t: eng.full((C,), a_post)  # standard-gamma draw replaced by its mean a
    out = smp.sample(state)
    eng.check_status()
    beta = state["parameter"].numpy()[:, :, 0]
    for c in range(C):
        r = state["response"][:, 0] - state["prefactor_matrix"] @ beta[c]
        assert np.allclose(1 / out["tau"].chain(c).item(), np.mean(r**2))
    eng.close()


def test_manifoldmala_recovers_gradient():
    """check_manifoldmala (tests/test_sampler.py:245-259): with z = 0 the proposal is the drift, and
    g = 2 H (x' - x) / step^2.  Forced acceptance (u -> 0) exposes the proposal."""
    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.engine import Engine
    from openmcmc_amd.model import Model
    from openmcmc_amd.sampler.metropolis_hastings import ManifoldMALA

    d, C, step = 6, 2, 0.3
    rng = np.random.default_rng(1)
    A = rng.standard_normal((d, 2 * d))
    Q = np.linalg.inv(A @ A.T / (2 * d))
    Q = (Q + Q.T) / 2
    mu = rng.standard_normal((d, 1))
    x0 = rng.standard_normal((C, d))
    eng = Engine(C)
    mdl = Model([Normal("parameter", mean="mu", precision="Q")])
    smp = ManifoldMALA("parameter", mdl, step=np.array([[step]])).bind(eng)
    smp.inject = lambda s_, t: eng.zeros(C, d)
    smp.inject_uniform = lambda s_, t: eng.full((C,), 1e-300)
    state = {"parameter": ChainArray(eng.to_device(x0)), "mu": mu, "Q": Q}
    out = smp.sample(state)
    eng.check_status()
    for c in range(C):
        grad = -Q @ (x0[c].reshape(d, 1) - mu)
        r = out["parameter"].chain(c) - x0[c].reshape(d, 1)
        assert np.allclose(grad, (Q @ r) * 2 / step**2, rtol=1e-5, atol=1e-8)
    assert smp.accept_rate.count == {"accept": C, "proposal": C}
    eng.close()


def test_run_mcmc_call_counts(golden):
    """tests/test_mcmc.py:83-124: sample() is called (n_iter + n_burn) * n_thin times per sampler,
    store() n_iter times, log_p n_iter times."""
    import test_mcmc_api_gpu as api

    G = golden("gmrf_chain")
    M, samplers = api.build(G, "sparse_", True, 2, fuse=False, n_burn=3, n_iter=4)
    M.n_thin = 2
    counts = {"sample": 0, "store": 0, "log_p": 0}
    for s in samplers:
        orig_sample, orig_store = s.sample, s.store
        s.sample = (lambda f: (lambda st: (counts.__setitem__("sample", counts["sample"] + 1), f(st))[1]))(orig_sample)
        s.store = (lambda f: (lambda **kw: (counts.__setitem__("store", counts["store"] + 1), f(**kw))[1]))(orig_store)
    orig_logp = M.model.log_p
    M.model.log_p = lambda *a, **k: (counts.__setitem__("log_p", counts["log_p"] + 1), orig_logp(*a, **k))[1]
    M.run_mcmc()
    assert counts["sample"] == (4 + 3) * 2 * len(samplers)
    assert counts["store"] == 4 * len(samplers)
    assert counts["log_p"] == 4
    assert not np.isnan(M.collect()["b"]).any()


@pytest.mark.parametrize("tag", ["tri", "band", "dense"])
def test_regression_under_a_correlated_response_replays_reference(golden, tag):
    """y ~ N(X beta, (tau W)^-1) with W tridiagonal, pentadiagonal or dense (sampler/sampler.py:179-192 and
    location_scale.py:190-250 accept any Q): NormalNormal(beta) + NormalGamma(tau) + NormalGamma(lambda) through MCMC.run_mcmc
    on the reference's recorded draws, its gradient / Hessian through the mean and its log density
    (tests/golden/correlated_regression.npz, made by tests/golden/make_golden_r4.py running the reference)."""
    from scipy import sparse

    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.distribution import Gamma
    from openmcmc_amd.distribution.location_scale import Normal, ScaledHessian
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix
    from openmcmc_amd.sampler.sampler import NormalGamma, NormalNormal

    G = golden("correlated_regression")
    k = tag + "_"
    N, p = int(G["N"]), int(G["p"])
    W = G[k + "W"] if tag == "dense" else sparse.csc_matrix(G[k + "W"])
    mdl = Model([Normal("y", mean=LinearCombination(form={"beta": "X"}), precision=ScaledMatrix(matrix="W", scalar="tau")),
                 Normal("beta", mean="mu_b", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
                 Gamma("tau", shape="a_tau", rate="b_tau"), Gamma("lambda", shape="a_lambda", rate="b_lambda")],
                response={"y": "mean"})
    state = {"y": G["y"], "X": G["X"], "beta": np.full(p, 0.1), "mu_b": np.full(p, 0.2), "W": W, "tau": 1.5,
             "P_lambda": sparse.identity(p, format="csc"), "lambda": 0.3, "a_tau": 1e-2, "b_tau": 1e-2, "a_lambda": 1e-2, "b_lambda": 1e-2}
    nn, g_tau, g_lam = NormalNormal("beta", mdl), NormalGamma("tau", mdl), NormalGamma("lambda", mdl)
    C = 3
    M = MCMC(state, [nn, g_tau, g_lam], model=mdl, n_burn=int(G["n_burn"]), n_iter=int(G["n_iter"]), n_chains=C)
    eng = M.engine
    # gradient through the mean and log density at the start state (before anything moves)
    grad, H = mdl["y"].grad_log_p(M.state, "beta", engine=eng)
    assert isinstance(H, ScaledHessian)
    for c in range(C):
        assert relerr(grad.data[c].cpu().numpy().ravel(), G[k + "grad"]) < 1e-11
    assert relerr(1.5 * np.asarray(H.matrix), G[k + "hess"]) < 1e-12 and relerr(H.scale.cpu().numpy(), np.full(C, 1.5)) < 1e-15
    assert relerr(mdl["y"].log_p(M.state, engine=eng).cpu().numpy(), np.full(C, G[k + "log_p"])) < 1e-11
    nn.inject = lambda s_, t: eng.to_device(np.tile(G[k + "z"][t], (C, 1)))
    g_tau.inject = lambda s_, t: eng.full((C,), G[k + "g"][t, 0])
    g_lam.inject = lambda s_, t: eng.full((C,), G[k + "g"][t, 1])
    M.run_mcmc()
    out = M.collect()
    for c in range(C):
        for key in ("beta", "tau", "lambda", "log_post", "y"):
            assert relerr(out[key][c], G[k + "store_" + key]) < 1e-9, (key, c)
    eng.close()


@pytest.mark.parametrize("tag", ["tri", "dense"])
def test_by_observation_of_a_fixed_size_replicated_response(golden, tag):
    """Normal.log_p(by_observation=True) (location_scale.py:145-167 -> gmrf.py:321-348): one log density per replicate column,
    for shared data under a per-chain mean and scalar, and for a per-chain replicated response; Gamma and Uniform alike."""
    from scipy import sparse, stats

    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.distribution import Gamma, Uniform
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.parameter import ScaledMatrix

    G = golden("correlated_regression")
    Y, mu, tau = G["byobs_Y"], G["byobs_mu"].reshape(-1, 1), float(G["byobs_tau"])
    d, n_rep = Y.shape
    W = G["byobs_" + tag + "_W"] if tag == "dense" else sparse.csc_matrix(G["byobs_" + tag + "_W"])
    want = G["byobs_" + tag + "_logp"]
    C = 4
    eng = make_engine(C)
    dist = Normal("Y", mean="mu", precision=ScaledMatrix(matrix="W", scalar="tau"))
    # (i) shared data, per-chain scalar (all chains at the reference's value) and per-chain mean
    st = {"Y": Y, "mu": ChainArray(eng.to_device(np.tile(mu.reshape(1, d, 1), (C, 1, 1)))), "W": W,
          "tau": ChainArray(eng.full((C, 1, 1), tau))}
    got = dist.log_p(st, by_observation=True, engine=eng).cpu().numpy()
    assert got.shape == (C, n_rep) and relerr(got, np.tile(want, (C, 1))) < 1e-12
    assert relerr(got.sum(axis=1), np.full(C, G["byobs_" + tag + "_total"])) < 1e-12
    # (ii) the response per chain (every chain its own shift), shared mean
    shift = np.arange(C, dtype=float).reshape(C, 1, 1) * 0.1
    st2 = {"Y": ChainArray(eng.to_device(Y[None] + shift)), "mu": mu, "W": W, "tau": ChainArray(eng.full((C, 1, 1), tau))}
    got2 = dist.log_p(st2, by_observation=True, engine=eng).cpu().numpy()
    Wd = W.toarray() if sparse.issparse(W) else W
    for c in range(C):
        r = Y + shift[c] - mu
        ref = 0.5 * (np.linalg.slogdet(tau * Wd)[1] - d * np.log(2 * np.pi) - np.sum(r * (tau * Wd @ r), axis=0))
        assert relerr(got2[c], ref) < 1e-12
    # Gamma / Uniform on a per-chain (p, n_rep) response (distribution.py:255-259, 436-440)
    xg = np.abs(Y) + 0.1
    gm = Gamma("g", shape="a", rate="b")
    gl = gm.log_p({"g": ChainArray(eng.to_device(np.tile(xg[None], (C, 1, 1)))), "a": 2.5, "b": 1.7}, by_observation=True, engine=eng).cpu().numpy()
    assert relerr(gl, np.tile(np.sum(stats.gamma.logpdf(xg, 2.5, scale=1 / 1.7), axis=0), (C, 1))) < 1e-12
    un = Uniform("u", domain_response_lower=np.full((d, 1), -4.0), domain_response_upper=np.full((d, 1), 5.0))
    ul = un.log_p({"u": ChainArray(eng.to_device(np.tile(Y[None], (C, 1, 1))))}, by_observation=True, engine=eng).cpu().numpy()
    assert relerr(ul, np.full((C, n_rep), -d * np.log(9.0))) < 1e-14
    eng.close()
