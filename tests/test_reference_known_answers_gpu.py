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
