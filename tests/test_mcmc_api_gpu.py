"""The reference-shaped API (Model / Normal / Gamma / ScaledMatrix / NormalNormal / NormalGamma /
MCMC.run_mcmc) on chain-batched GPU state, replayed against the reference's own output
(tests/golden/gmrf_chain.npz) and checked for fused == unfused."""

import numpy as np
import pytest
from scipy import sparse

pytestmark = pytest.mark.gpu
TOL = 1e-10


def relerr(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def build(G, k, sparse_route, n_chains, fuse, n_burn=None, n_iter=None, seed=0, **mcmc_kw):
    from openmcmc_amd.distribution.distribution import Gamma
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix
    from openmcmc_amd.sampler.sampler import NormalGamma, NormalNormal

    n = int(G[k + "n"])
    pd, po = G[k + "P_diag"], G[k + "P_off"]
    P = sparse.diags((po, pd, po), offsets=[-1, 0, 1], format="csc")
    mean = LinearCombination(form={"b": "A"}) if sparse_route else "b"
    mdl = Model([
        Normal("y", mean=mean, precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
        Normal("b", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
        Gamma("lambda", shape="a_lam", rate="b_lam"),
        Gamma("tau", shape="a_tau", rate="b_tau"),
    ])
    state = {"y": G[k + "y"], "b": G[k + "y"], "mu": np.full(n, float(G[k + "mu_val"])), "lambda": 100,
             "P_lambda": P, "a_lam": 10, "b_lam": 1, "tau": 1, "P_tau": sparse.csc_matrix(np.eye(n)),
             "a_tau": 1, "b_tau": 1}
    if sparse_route:
        state["A"] = sparse.identity(n, format="csc")
    samplers = [NormalNormal("b", mdl), NormalGamma("lambda", mdl), NormalGamma("tau", mdl)]
    M = MCMC(state, samplers, model=mdl, n_burn=int(G[k + "n_burn"]) if n_burn is None else n_burn,
             n_iter=int(G[k + "n_iter"]) if n_iter is None else n_iter, n_chains=n_chains, fuse=fuse, seed=seed, **mcmc_kw)
    return M, samplers


@pytest.mark.parametrize("route", ["sparse", "dense", "sparsemu"])
def test_run_mcmc_replays_reference(golden, route):
    """Same model objects, same sampler list, the reference's recorded draws injected through the
    samplers' `inject` hook: store matches the reference's store for every chain."""
    G = golden("gmrf_chain")
    k = route + "_"
    C = 3
    M, (nn, g_lam, g_tau) = build(G, k, route != "dense", C, fuse=False)
    eng = M.engine
    nn.inject = lambda smp, t: eng.to_device(np.tile(G[k + "z"][t], (C, 1)))
    g_lam.inject = lambda smp, t: eng.full((C,), G[k + "g"][t, 0])
    g_tau.inject = lambda smp, t: eng.full((C,), G[k + "g"][t, 1])
    M.run_mcmc()
    out = M.collect()
    assert out["b"].shape == (C,) + G[k + "store_b"].shape
    for c in range(C):
        assert relerr(out["b"][c], G[k + "store_b"]) < TOL
        assert relerr(out["lambda"][c], G[k + "store_lambda"]) < TOL
        assert relerr(out["tau"][c], G[k + "store_tau"]) < TOL
        assert relerr(out["log_post"][c], G[k + "store_log_post"]) < TOL


def test_fused_and_unfused_loops_agree(golden):
    """fuse=True hands the whole sampler list to omc_gmrf_sweep; results equal the sampler-by-sampler
    loop to rounding (same random streams; the unfused NormalGamma sums its quadratic form in a
    different order)."""
    G = golden("gmrf_chain")
    outs = []
    for fuse in (False, True):
        M, _ = build(G, "sparse_", True, 5, fuse=fuse, n_burn=4, n_iter=9, seed=123)
        M.n_thin = 2  # burn-in and stored iterations are both n_thin sweeps long (mcmc.py:97-98)
        assert (M._fused is not None) == fuse
        M.run_mcmc()
        outs.append(M.collect())
    for key in ("b", "lambda", "tau", "log_post"):
        assert relerr(outs[1][key], outs[0][key]) < 1e-11, key
    # chains are distinct draws from the same posterior
    assert not np.array_equal(outs[0]["b"][0], outs[0]["b"][1])


def test_store_shapes_and_state_untouched(golden):
    """mcmc.py:78-85 / tests/test_sampler.py:181-198 of the reference: shapes, NaN-initialised store,
    and no state entry other than the sampled one changes."""
    G = golden("gmrf_chain")
    M, (nn, g_lam, g_tau) = build(G, "sparse_", True, 2, fuse=False, n_burn=0, n_iter=3)
    n = int(G["sparse_n"])
    assert tuple(M.store["b"].shape) == (3, 2, n) and tuple(M.store["log_post"].shape) == (3, 2)
    assert np.isnan(M.store["b"].cpu().numpy()).all()
    before = {key: (v.numpy().copy() if hasattr(v, "numpy") and not isinstance(v, np.ndarray) else v)
              for key, v in M.state.items()}
    M.state = nn.sample(M.state)
    assert M.state["b"].shape == (n, 1)
    for key in ("lambda", "tau"):
        assert np.array_equal(M.state[key].numpy(), before[key])
    M.state = g_lam.sample(M.state)
    assert not np.array_equal(M.state["lambda"].numpy(), before["lambda"])
    assert np.array_equal(M.state["tau"].numpy(), before["tau"])


def test_dense_prior_and_unsupported_structures():
    """A dense prior precision goes down the dense path (and matches the oracle); structures that are
    not built yet fail loudly instead of falling back to anything."""
    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.engine import Engine
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import ScaledMatrix
    from openmcmc_amd.sampler.sampler import NormalNormal
    from oracle import gmrf_ref

    n = 6
    rng = np.random.default_rng(2)
    dense = np.eye(n) + 0.1 * np.ones((n, n))
    yv = rng.standard_normal((n, 1))
    mdl = Model([Normal("y", mean="b", precision=ScaledMatrix("P_tau", "tau")),
                 Normal("b", mean="mu", precision=ScaledMatrix("P", "lam"))])
    eng = Engine(2)
    smp = NormalNormal("b", mdl).bind(eng)
    state = {"y": yv, "b": ChainArray(eng.zeros(2, n)), "mu": np.zeros((n, 1)), "P": dense,
             "lam": 1.5, "P_tau": sparse.identity(n, format="csc"), "tau": 2.0}
    z = rng.standard_normal(n)
    smp.inject = lambda s_, t: eng.to_device(np.tile(z, (2, 1)))
    state = smp.sample(state)
    eng.check_status()
    xo, _, _ = gmrf_ref.draw_canonical(2.0 * yv, 1.5 * dense + 2.0 * np.eye(n), z)
    assert relerr(state["b"].chain(1).ravel(), xo.ravel()) < TOL
    with pytest.raises(RuntimeError):
        NormalNormal("b", mdl).sample(state)  # not bound to an engine
    # a truncated prior on the dense route: one scan of single-site truncated updates, checked against the oracle
    trunc = Model([Normal("y", mean="b", precision=ScaledMatrix("P_tau", "tau")),
                   Normal("b", mean="mu", precision=ScaledMatrix("P", "lam"), domain_response_lower=np.zeros((n, 1)))])
    tsmp = NormalNormal("b", trunc).bind(eng)
    u = rng.random(n)
    tsmp.inject = lambda s_, t: eng.to_device(np.tile(u, (2, 1)))
    x0 = np.abs(state["b"].chain(1)) + 0.1
    state["b"] = ChainArray(eng.to_device(np.tile(x0.reshape(1, n), (2, 1))))
    state = tsmp.sample(state)
    eng.check_status()
    xt = gmrf_ref.gibbs_truncated_scan(2.0 * yv, 1.5 * dense + 2.0 * np.eye(n), x0, 0.0, None, u)
    assert relerr(state["b"].chain(0).ravel(), xt.ravel()) < TOL and state["b"].chain(0).min() >= 0.0
    with pytest.raises(ValueError):  # gmrf.py:149-150
        bad = Model([Normal("y", mean="b", precision=ScaledMatrix("P_tau", "tau")),
                     Normal("b", mean="mu", precision=ScaledMatrix("P", "lam"), domain_response_lower=np.array(1.0),
                            domain_response_upper=np.array(0.5))])
        NormalNormal("b", bad).bind(eng).sample(state)
    eng.close()


def build_linreg(G, k, C, **mcmc_kw):
    """examples/3_linear_regression.ipynb's model and sampler list on C chains (in-kernel draws unless the caller injects)."""
    from openmcmc_amd.distribution.distribution import Gamma
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix
    from openmcmc_amd.sampler.sampler import NormalGamma, NormalNormal

    X, y = G[k + "X"], G[k + "y"]
    N, p = X.shape
    mdl = Model([Normal("y", mean=LinearCombination(form={"beta": "X"}), precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
                 Normal("beta", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
                 Gamma("tau", shape="a_tau", rate="b_tau"),
                 Gamma("lambda", shape="a_lambda", rate="b_lambda")], response={"y": "mean"})
    samplers = [NormalNormal("beta", mdl), NormalGamma("tau", mdl), NormalGamma("lambda", mdl)]
    state = {"y": y, "X": X, "beta": [0.0] * p, "P_tau": sparse.csc_matrix(np.eye(N)), "tau": 1,
             "P_lambda": sparse.csc_matrix(np.eye(p)), "mu": [0.0] * p, "lambda": 0.01,
             "a_tau": 1e-3, "b_tau": 1e-3, "a_lambda": 1e-3, "b_lambda": 1e-3}
    return MCMC(state, samplers, model=mdl, n_burn=int(G[k + "n_burn"]), n_iter=int(G[k + "n_iter"]), n_chains=C, seed=13, **mcmc_kw)


@pytest.mark.parametrize("tag", ["ex3", "p7"])
def test_linear_regression_example_replays_reference(golden, tag):
    """BASELINE configs[0]: examples/3_linear_regression.ipynb verbatim (model, samplers, state,
    response={'y': 'mean'}) on the chain-batched API; the reference's recorded draws injected."""
    from openmcmc_amd.distribution.distribution import Gamma
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix
    from openmcmc_amd.sampler.sampler import NormalGamma, NormalNormal

    G = golden("linreg_chain")
    k = tag + "_"
    X, y = G[k + "X"], G[k + "y"]
    N, p = X.shape
    mdl = Model([Normal("y", mean=LinearCombination(form={"beta": "X"}), precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
                 Normal("beta", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
                 Gamma("tau", shape="a_tau", rate="b_tau"),
                 Gamma("lambda", shape="a_lambda", rate="b_lambda")], response={"y": "mean"})
    samplers = [NormalNormal("beta", mdl), NormalGamma("tau", mdl), NormalGamma("lambda", mdl)]
    state = {"y": y, "X": X, "beta": [0.0] * p, "P_tau": sparse.csc_matrix(np.eye(N)), "tau": 1,
             "P_lambda": sparse.csc_matrix(np.eye(p)), "mu": [0.0] * p, "lambda": 0.01,
             "a_tau": 1e-3, "b_tau": 1e-3, "a_lambda": 1e-3, "b_lambda": 1e-3}
    C = 2
    M = MCMC(state, samplers, model=mdl, n_burn=int(G[k + "n_burn"]), n_iter=int(G[k + "n_iter"]), n_chains=C)
    assert M._fused is None  # dense conditional: sampler-by-sampler loop
    eng = M.engine
    samplers[0].inject = lambda smp, t: eng.to_device(np.tile(G[k + "z"][t], (C, 1)))
    samplers[1].inject = lambda smp, t: eng.full((C,), G[k + "g"][t, 0])
    samplers[2].inject = lambda smp, t: eng.full((C,), G[k + "g"][t, 1])
    M.run_mcmc()
    out = M.collect()
    for c in range(C):
        for key in ("beta", "tau", "lambda", "log_post", "y"):
            assert relerr(out[key][c], G[k + "store_" + key]) < 1e-9, key


@pytest.mark.parametrize("kind", ["mala", "rw"])
def test_mh_samplers_replay_reference(golden, kind):
    """ManifoldMALA / RandomWalk objects with the reference's constructor signature on the cfg4 model
    Normal("x", mean="mu", precision="Q"): 40-step traces of the reference, accept counts bit-exact."""
    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.engine import Engine
    from openmcmc_amd.model import Model
    from openmcmc_amd.sampler.metropolis_hastings import ManifoldMALA, RandomWalk

    G = golden("mala")
    d, C = 32, 2
    k = f"d{d}_"
    mdl = Model([Normal("x", mean="mu", precision="Q")])
    eng = Engine(C)
    cls = ManifoldMALA if kind == "mala" else RandomWalk
    smp = cls("x", mdl, step=np.array([[float(G[k + kind + "_step"])]])).bind(eng)
    state = {"x": ChainArray(eng.to_device(np.tile(G[k + "x0"], (C, 1)))), "mu": np.zeros((d, 1)), "Q": G[k + "Q"]}
    smp.inject = lambda s_, t: eng.to_device(np.tile(G[k + kind + "_z"][t], (C, 1)))
    smp.inject_uniform = lambda s_, t: eng.full((C,), G[k + kind + "_u"][t])
    assert smp.accept_rate.get_acceptance_rate() == "No proposals"
    for i in range(G[k + kind + "_z"].shape[0]):
        state = smp.sample(state)
        assert relerr(state["x"].chain(1).ravel(), G[k + kind + "_x"][i]) < 1e-9
    n_acc = int(G[k + kind + "_accept"].sum())
    assert smp.accept_rate.count == {"accept": C * n_acc, "proposal": C * 40}
    assert smp.accept_rate.get_acceptance_rate() == f"Acceptance rate {100 * n_acc / 40:.0f}%"
    eng.close()


def test_grad_log_p_and_prior_draws(golden):
    """Model.grad_log_p on the cfg4 model equals the reference's gradient/Hessian (mala.npz); Normal.rvs
    starts chains from the prior when the state has no value (mcmc.py:78-80): Mahalanobis ~ chi^2_d."""
    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.engine import Engine
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import ScaledMatrix

    G = golden("mala")
    d, C = 32, 3
    k = f"d{d}_"
    eng = Engine(C, seed=8)
    mdl = Model([Normal("x", mean="mu", precision="Q")])
    state = {"x": ChainArray(eng.to_device(np.tile(G[k + "x0"], (C, 1)))), "mu": np.zeros((d, 1)), "Q": G[k + "Q"]}
    grad, hess = mdl.grad_log_p(state, "x", engine=eng)
    assert relerr(grad.chain(2).ravel(), G[k + "one_grad"]) < TOL and np.array_equal(hess, G[k + "one_hess"])
    eng.close()
    # prior draws: dense and tridiagonal
    C = 4000
    eng = Engine(C, seed=9)
    Q = G[k + "Q"]
    x = Normal("x", mean="mu", precision="Q").rvs({"mu": np.ones((d, 1)), "Q": Q}, engine=eng, draw_index=3)
    r = x.numpy()[:, :, 0] - 1.0
    maha = np.einsum("ci,ij,cj->c", r, Q, r)
    assert abs(maha.mean() / d - 1) < 0.05 and abs(maha.var() / (2 * d) - 1) < 0.15
    n = 50
    P = sparse.diags((-np.ones(n - 1), 2.5 * np.ones(n), -np.ones(n - 1)), offsets=[-1, 0, 1], format="csc")
    dist = Normal("b", mean="m", precision=ScaledMatrix("P", "lam"))
    xb = dist.rvs({"m": np.full((n, 1), 0.3), "P": P, "lam": np.array([[4.0]])}, engine=eng, draw_index=4)
    rb = xb.numpy()[:, :, 0] - 0.3
    maha = 4.0 * np.einsum("ci,ij,cj->c", rb, P.toarray(), rb)
    assert abs(maha.mean() / n - 1) < 0.05
    eng.check_status()
    eng.close()


def test_on_device_posterior_summaries(golden):
    """MCMC.summary: mean/variance of the device-resident store without gathering it."""
    G = golden("gmrf_chain")
    M, _ = build(G, "sparse_", True, 6, fuse=True, n_burn=3, n_iter=12, seed=5)
    M.run_mcmc()
    out = M.collect()
    for key in ("b", "lambda", "log_post"):
        arr = out[key] if key != "log_post" else np.transpose(out[key], (0, 2, 1))  # (C, size, n_iter)
        mean, var = M.summary(key, pooled=False)
        assert relerr(mean, arr.mean(axis=2)) < 1e-12 and relerr(var, arr.var(axis=2, ddof=1)) < 1e-10
        mean, var = M.summary(key, pooled=True)
        flat = np.transpose(arr, (1, 0, 2)).reshape(arr.shape[1], -1)
        assert relerr(mean, flat.mean(axis=1)) < 1e-12 and relerr(var, flat.var(axis=1, ddof=1)) < 1e-10


def test_example2_samplers_replay_reference(golden):
    """examples/2_samplers.ipynb verbatim through the mirror API: a scalar h observed five times (y is (1, 5):
    replicated response), RandomWalk('h', step=5.0) on the generic Metropolis-Hastings route and
    NormalNormal('h'); 300 iterations each with the reference's recorded draws."""
    import torch

    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.sampler.metropolis_hastings import RandomWalk
    from openmcmc_amd.sampler.sampler import NormalNormal

    G = golden("example2")
    n_iter = int(G["n_iter"])

    def fresh():
        mdl = Model([Normal("y", mean="h", precision="tau"), Normal("h", mean="mu", precision="lambda")])
        st = {"y": np.array(G["y"], ndmin=2), "h": np.array(float(G["h0"]), ndmin=2), "tau": np.array(float(G["tau"]), ndmin=2),
              "mu": np.array(float(G["mu"]), ndmin=2), "lambda": np.array(float(G["lambda"]), ndmin=2)}
        return mdl, st

    C = 3  # three identical chains fed the same draws: every one must reproduce the reference
    dev = torch.device("cuda", 0)
    mdl, st = fresh()
    smp = RandomWalk("h", model=mdl, step=float(G["step"]))
    smp.inject = lambda s, it, j: torch.full((C, 1), float(G["rw_z"][it]), dtype=torch.float64, device=dev)
    smp.inject_uniform = lambda s, it, j: torch.full((C,), float(G["rw_u"][it]), dtype=torch.float64, device=dev)
    M = MCMC(st, [smp], model=mdl, n_burn=0, n_iter=n_iter, n_chains=C)
    M.run_mcmc()
    got = M.collect()
    for c in range(C):
        assert np.max(np.abs(got["h"][c] - G["rw_store_h"]) / np.abs(G["rw_store_h"])) < 1e-12
        assert np.max(np.abs(got["log_post"][c] - G["rw_log_post"]) / np.abs(G["rw_log_post"])) < 1e-10
    assert smp.accept_rate.accept.cpu().numpy().tolist() == [int(G["rw_accept"][0])] * C
    assert smp.accept_rate.proposal.cpu().numpy().tolist() == [int(G["rw_accept"][1])] * C

    mdl, st = fresh()
    nn = NormalNormal("h", model=mdl)
    nn.inject = lambda s, it: torch.full((C, 1), float(G["nn_z"][it]), dtype=torch.float64, device=dev)
    M = MCMC(st, [nn], model=mdl, n_burn=0, n_iter=n_iter, n_chains=C)
    M.run_mcmc()
    got = M.collect()
    for c in range(C):
        assert np.max(np.abs(got["h"][c] - G["nn_store_h"]) / np.abs(G["nn_store_h"])) < 1e-10
        assert np.max(np.abs(got["log_post"][c] - G["nn_log_post"]) / np.abs(G["nn_log_post"])) < 1e-10


def test_dense_prior_normal_gamma_and_log_p():
    """A DENSE (not banded) prior precision: NormalGamma's r'Mr through one GEMM + omc_centered_rowdot and the
    log-determinant of M by one device Cholesky, against the oracle's restatement of sampler.py:276-287 and
    gmrf.py:321-348."""
    from scipy import sparse

    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.distribution import Gamma
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.engine import Engine
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import ScaledMatrix
    from openmcmc_amd.sampler.sampler import NormalGamma
    from oracle import gmrf_ref, sweep_ref

    p, C = 30, 4
    rng = np.random.default_rng(21)
    A = rng.standard_normal((p, 2 * p))
    Qd = A @ A.T / (2 * p) + 0.3 * np.eye(p)
    Qd = (Qd + Qd.T) / 2
    mu = rng.standard_normal((p, 1)) * 0.5
    mdl = Model([Normal("beta", mean="mu", precision=ScaledMatrix("Q", "lam")), Gamma("lam", shape="a", rate="b")])
    eng = Engine(C)
    beta = rng.standard_normal((C, p))
    lam0 = rng.random(C) + 0.5
    state = {"beta": ChainArray(eng.to_device(beta)), "mu": mu, "Q": Qd, "a": np.array([[2.0]]), "b": np.array([[1.5]]),
             "lam": ChainArray(eng.to_device(lam0).reshape(C, 1, 1))}
    lp = mdl.log_p(state, engine=eng).cpu().numpy()
    for c in range(C):
        ref = gmrf_ref.gauss_logpdf(beta[c].reshape(p, 1), mu, lam0[c] * Qd) + sweep_ref.gamma_logpdf(lam0[c], 2.0, 1.5)
        assert abs(lp[c] - ref) < 1e-10 * max(1.0, abs(ref))
    g = rng.gamma(2.0 + p / 2, size=C)
    smp = NormalGamma("lam", mdl).bind(eng)
    smp.inject = lambda s_, t: eng.to_device(g)
    state = smp.sample(state)
    eng.check_status()
    got = state["lam"].scalar().cpu().numpy()
    for c in range(C):
        a_c, b_c = sweep_ref.gamma_conditional(2.0, 1.5, beta[c].reshape(p, 1) - mu, sparse.csc_matrix(Qd))
        assert abs(got[c] - sweep_ref.gamma_draw_from_standard(a_c, b_c, g[c])) < 1e-10 * got[c]
    eng.close()


def test_mixture_prior_model_replays_reference(golden):
    """The mixture-prior model of the reference's sampler tests: NormalNormal on a parameter vector whose prior mean and
    precision are picked by a categorical allocation (dense route + per-chain diagonal), NormalGamma on the
    per-component precisions, MixtureAllocation; Categorical / vector-Gamma terms in log_post.  40 sweeps with the
    reference's draws (tests/golden/mixture_chain.npz); allocations must be identical, the rest to 1e-10."""
    import torch

    from openmcmc_amd.distribution.distribution import Categorical, Gamma
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import Identity, LinearCombination, MixtureParameterMatrix, MixtureParameterVector
    from openmcmc_amd.sampler.sampler import MixtureAllocation, NormalGamma, NormalNormal

    G = golden("mixture_chain")
    n, p, K, n_iter = int(G["n"]), int(G["p"]), int(G["K"]), int(G["n_iter"])
    st = {"response": G["y"].reshape(n, 1), "prefactor_matrix": G["X"], "parameter": G["parameter0"].reshape(p, 1),
          "prior_mean": G["prior_mean"].reshape(K, 1), "precision_matrix": np.diag(G["w"]), "prior_precision_vector": G["prec0"],
          "gamma_shape": 2.0 * np.ones((K,)), "gamma_rate": 1.0 * np.ones((K,)), "allocation": G["alloc0"].reshape(p, 1),
          "prior_allocation_prob": G["prob"]}
    mdl = Model([
        Normal("response", mean=LinearCombination({"parameter": "prefactor_matrix"}), precision=Identity("precision_matrix")),
        Normal("parameter", mean=MixtureParameterVector("prior_mean", "allocation"),
               precision=MixtureParameterMatrix("prior_precision_vector", "allocation")),
        Gamma("prior_precision_vector", shape=Identity("gamma_shape"), rate=Identity("gamma_rate")),
        Categorical("allocation", prob="prior_allocation_prob")])
    samplers = [NormalNormal("parameter", mdl), NormalGamma("prior_precision_vector", mdl),
                MixtureAllocation("allocation", mdl, response_param="parameter")]
    C = 2
    dev = torch.device("cuda", 0)
    tile = lambda a: torch.as_tensor(np.tile(a, (C, 1)), device=dev)  # noqa: E731
    samplers[0].inject = lambda s, it: tile(G["z"][it])
    samplers[1].inject = lambda s, it: tile(G["g"][it])
    samplers[2].inject = lambda s, it: tile(G["u"][it])
    M = MCMC(st, samplers, model=mdl, n_burn=0, n_iter=n_iter, n_chains=C)
    M.run_mcmc()
    got = M.collect()
    for c in range(C):
        assert np.array_equal(got["allocation"][c], G["store_allocation"])
        for key in ("parameter", "prior_precision_vector", "log_post"):
            ref = G["store_" + key]
            err = np.max(np.abs(got[key][c] - ref) / np.maximum(1.0, np.abs(ref)))
            assert err < 1e-10, (key, err)


def test_lognormal_and_categorical_log_p():
    """LogNormal.log_p (location_scale.py:279-300) for a per-chain response and for a shared response with a per-chain
    mean, against scipy.stats.lognorm; Categorical.log_p (distribution.py:318-352) and rvs frequencies."""
    from scipy import sparse, stats

    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.distribution import Categorical
    from openmcmc_amd.distribution.location_scale import LogNormal
    from openmcmc_amd.engine import Engine
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import ScaledMatrix

    n, C = 9, 5
    rng = np.random.default_rng(2)
    eng = Engine(C, seed=3)
    d = LogNormal("y", mean="m", precision=ScaledMatrix("P", "tau"))
    tau = rng.random(C) + 0.5
    y = np.exp(rng.standard_normal((C, n)))
    m = rng.standard_normal((n, 1))
    state = {"y": ChainArray(eng.to_device(y)), "m": m, "P": sparse.identity(n, format="csc"),
             "tau": ChainArray(eng.to_device(tau).reshape(C, 1, 1))}
    lp = Model([d]).log_p(state, engine=eng).cpu().numpy()
    for c in range(C):
        ref = stats.lognorm.logpdf(y[c], s=1 / np.sqrt(tau[c]), scale=np.exp(m.ravel())).sum()
        assert abs(lp[c] - ref) < 1e-10 * max(1.0, abs(ref))
    # shared response, per-chain mean
    ys = np.exp(rng.standard_normal((n, 1)))
    mc = rng.standard_normal((C, n))
    state = {"y": ys, "m": ChainArray(eng.to_device(mc)), "P": sparse.identity(n, format="csc"),
             "tau": ChainArray(eng.to_device(tau).reshape(C, 1, 1))}
    lp = Model([d]).log_p(state, engine=eng).cpu().numpy()
    for c in range(C):
        ref = stats.lognorm.logpdf(ys.ravel(), s=1 / np.sqrt(tau[c]), scale=np.exp(mc[c])).sum()
        assert abs(lp[c] - ref) < 1e-10 * max(1.0, abs(ref))
    # Categorical
    prob = np.array([[0.2, 0.5, 0.3]])
    cat = Categorical("z", prob="pi")
    z = rng.integers(0, 3, size=(C, 6)).astype(float)
    st = {"z": ChainArray(eng.to_device(z)), "pi": prob}
    lp = Model([cat]).log_p(st, engine=eng).cpu().numpy()
    assert np.allclose(lp, np.log(prob[0][z.astype(int)]).sum(axis=1), rtol=1e-13)
    assert cat.log_p({"z": z[0].reshape(6, 1), "pi": prob}) == pytest.approx(lp[0])
    eng2 = Engine(20000, seed=5)
    draws = cat.rvs({"z": np.zeros((1, 1)), "pi": prob}, engine=eng2).data.cpu().numpy().ravel()
    freq = np.bincount(draws.astype(int), minlength=3) / draws.size
    assert np.all(np.abs(freq - prob[0]) < 0.01)
    eng.check_status()
    eng.close(), eng2.close()


def test_gradient_branches_match_reference(golden):
    """Normal.grad_log_p in its three branches and the finite-difference default of the Distribution base class
    (location_scale.py:190-250, distribution.py:90-198) against the reference (tests/golden/gradients.npz), three chains
    with different coefficients and precisions evaluated together."""
    from scipy import sparse

    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.distribution import Gamma
    from openmcmc_amd.distribution.location_scale import Normal, ScaledHessian
    from openmcmc_amd.engine import Engine
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix

    G = golden("gradients")
    X, w, y = G["X"], G["w"], G["y"]
    n, p = X.shape
    C = 3
    eng = Engine(C)
    lik = Normal("y", mean=LinearCombination({"beta": "X"}), precision=ScaledMatrix("P_tau", "tau"))
    prior = Gamma("tau", shape="a", rate="b")
    state = {"y": y.reshape(n, 1), "X": X, "beta": ChainArray(eng.to_device(G["beta"])), "P_tau": sparse.diags(w, format="csc"),
             "tau": ChainArray(eng.to_device(G["tau"]).reshape(C, 1, 1)), "a": np.array([[float(G["a"])]]), "b": np.array([[float(G["b"])]])}
    grad, hess = lik.grad_log_p(state, "beta", engine=eng)                      # (ii)
    assert isinstance(hess, ScaledHessian)
    for c in range(C):
        ref_g, ref_h = G[f"c{c}_grad_beta"], G[f"c{c}_hess_beta"]
        assert np.max(np.abs(grad.chain(c).ravel() - ref_g)) < 1e-10 * np.abs(ref_g).max()
        got_h = hess.scale[c].item() * hess.matrix
        assert np.max(np.abs(got_h - ref_h)) < 1e-10 * np.abs(ref_h).max()
    g_tau, h_tau = lik.grad_log_p(state, "tau", engine=eng)                     # (iii) finite differences
    gp, hp = prior.grad_log_p(state, "tau", engine=eng)                         # base-class default
    for c in range(C):
        # differences of log densities of size ~1e2 over a step of 1e-4: rounding of the densities (1e-13 relative)
        # is amplified by 1e4 / |gradient|, so these are compared at 1e-6 (the reference's own numbers carry the same noise)
        assert abs(g_tau.chain(c).item() - float(G[f"c{c}_grad_tau_lik"])) < 1e-6 * abs(float(G[f"c{c}_grad_tau_lik"]))
        assert abs(h_tau[c, 0, 0].item() - float(G[f"c{c}_hess_tau_lik"])) < 1e-4 * abs(float(G[f"c{c}_hess_tau_lik"]))
        assert abs(gp.chain(c).item() - float(G[f"c{c}_grad_tau_prior"])) < 1e-6 * max(1.0, abs(float(G[f"c{c}_grad_tau_prior"])))
        assert abs(hp[c, 0, 0].item() - float(G[f"c{c}_hess_tau_prior"])) < 1e-4 * abs(float(G[f"c{c}_hess_tau_prior"]))
    eng.close()


@pytest.mark.parametrize("kind", ["mala", "rw"])
def test_log_post_of_the_fused_mh_steps_is_the_models_log_p(kind):
    """One MH sampler on a one-Normal model: the whitened steps leave the target's log density of the state they end in
    (omc_*_step_white log_p_out) and MCMC.run_mcmc stores that instead of evaluating Model.log_p again (mcmc.py:99-111).
    Checked against the density evaluated in numpy at every stored state, and against the generic evaluation."""
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.sampler.metropolis_hastings import ManifoldMALA, RandomWalk

    d, C = 24, 5
    rng = np.random.default_rng(3)
    A = rng.standard_normal((d, 2 * d))
    Q = np.linalg.inv(A @ A.T / (2 * d))
    Q = (Q + Q.T) / 2
    mu = rng.standard_normal((d, 1))
    out = {}
    for fused_lp in (True, False):
        mdl = Model([Normal("x", mean="mu", precision="Q")])
        cls = ManifoldMALA if kind == "mala" else RandomWalk
        smp = cls("x", mdl, step=np.array([[0.6 if kind == "mala" else 0.2]]))
        M = MCMC({"x": np.zeros(d), "mu": mu, "Q": Q}, [smp], model=mdl, n_burn=5, n_iter=40, n_chains=C, seed=11)
        if not fused_lp:  # force the generic evaluation: hide the step's by-product
            inner = smp.sample

            def sample(state, inner=inner, smp=smp):
                state = inner(state)
                smp.last_log_p = None
                return state

            smp.sample = sample
        M.run_mcmc()
        out[fused_lp] = M.collect()
    x, lp = out[True]["x"], out[True]["log_post"]
    _, logdet = np.linalg.slogdet(Q)
    for c in range(C):
        r = x[c] - mu  # (d, n_iter)
        ref = 0.5 * (logdet - d * np.log(2 * np.pi) - np.einsum("it,ij,jt->t", r, Q, r))
        assert np.max(np.abs(lp[c].ravel() - ref)) < 1e-9 * max(1.0, np.abs(ref).max())
    assert np.array_equal(out[True]["x"], out[False]["x"])
    assert np.max(np.abs(out[True]["log_post"] - out[False]["log_post"])) < 1e-9 * np.abs(out[False]["log_post"]).max()


@pytest.mark.parametrize("ring", [0, 10])
def test_run_mcmc_hands_manifold_mala_whole_blocks(ring):
    """MCMC.run_mcmc on the cfg4 model with ManifoldMALA alone: the loop goes to the library in blocks (omc_mala_run_white);
    store, log_post, counters and final state are those of the loop that calls sample() once per iteration -- also with a
    ring store drained while it runs."""
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.sampler.metropolis_hastings import ManifoldMALA

    d, C, n_burn, n_iter = 40, 6, 37, 45
    rng = np.random.default_rng(8)
    A = rng.standard_normal((d, 2 * d))
    Qh = np.linalg.inv(A @ A.T / (2 * d))
    Qh = (Qh + Qh.T) / 2
    mu = rng.standard_normal((d, 1))
    x0 = mu.T + np.linalg.solve(np.linalg.cholesky(Qh).T, rng.standard_normal((d, C))).T
    outs = []
    for blocks in (False, True):
        mdl = Model([Normal("x", mean="mu", precision="Q")])
        smp = ManifoldMALA("x", mdl, step=np.array([[0.6]]))
        from openmcmc_amd.chains import ChainArray
        from openmcmc_amd.engine import Engine

        eng = Engine(C, seed=12)
        state = {"x": ChainArray(eng.to_device(x0[:, :, None])), "mu": mu, "Q": Qh}
        M = MCMC(state, [smp], model=mdl, n_burn=n_burn, n_iter=n_iter, n_chains=C, engine=eng, store_ring=ring if blocks else 0)
        if not blocks:
            smp.can_run_block = lambda st: False  # the loop of single steps
        assert M._mala_block_route() == blocks
        M.run_mcmc()
        outs.append((M.collect(), smp.accept_rate.accept.cpu().numpy().copy(), smp.accept_rate.proposal.cpu().numpy().copy(),
                     M.state["x"].data.cpu().numpy().copy()))
        eng.close()
    (s0, a0, p0, x_0), (s1, a1, p1, x_1) = outs
    assert np.array_equal(a0, a1) and np.array_equal(p0, p1) and p0[0] == n_burn + n_iter and 0 < a0.sum() < p0.sum()
    assert np.array_equal(s0["log_post"], s1["log_post"])
    assert relerr(s1["x"], s0["x"]) < 1e-12 and relerr(x_1, x_0) < 1e-12
