"""Truncated Gaussian full conditional on the GPU: omc_tridiag_gibbs_truncated /
omc_dense_gibbs_truncated against gmrf.gibbs_canonical_truncated_normal of the reference (golden vectors with the
recorded uniforms), NormalNormal with a truncated prior through MCMC.run_mcmc, and the -inf domain rule of
Normal.log_p."""

import numpy as np
import pytest
from scipy import sparse

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def make_engine(C, **kw):
    from openmcmc_amd.engine import Engine

    return Engine(C, **kw)


def test_gibbs_scan_matches_reference(golden):
    G = golden("truncated_conditional")
    C = 3
    eng = make_engine(C)
    worst = 0.0
    for ci in range(int(G["n_cases"])):
        k = f"c{ci}_"
        n, Q, b = int(G[k + "n"]), G[k + "Q"], G[k + "b"]
        lo, hi = float(G[k + "lower"]), float(G[k + "upper"])
        lower = None if np.isneginf(lo) else eng.full((n,), lo)
        upper = None if np.isposinf(hi) else eng.full((n,), hi)
        x = eng.to_device(np.tile(G[k + "x0"], (C, 1)))
        u = eng.to_device(np.tile(G[k + "u"], (C, 1)))
        scale = eng.full((C,), 2.0)  # Q = 2 * (Q / 2): exercises the per-chain scalar
        if str(G[k + "kind"]) == "tri":
            terms = [{"diag": eng.to_device(np.diag(Q) / 2), "off": eng.to_device(np.diag(Q, 1) / 2) if n > 1 else None,
                      "rhs": eng.to_device(b / 2), "scale": scale}]
            eng.tridiag_gibbs_truncated(n, terms, x, lower=lower, upper=upper, u=u)
        else:
            terms = [{"mat": eng.to_device(Q / 2), "rhs": eng.to_device(b / 2), "scale": scale}]
            eng.dense_gibbs_truncated(n, terms, x, lower=lower, upper=upper, u=u)
        eng.check_status()
        got = x.cpu().numpy()
        ref = G[k + "x"]
        err = np.max(np.abs(got - ref[None, :]) / np.maximum(1e-3, np.abs(ref[None, :])))
        worst = max(worst, err)
        assert err < RTOL, (ci, str(G[k + "kind"]), n, lo, hi, err)
        assert np.all(got >= lo) and np.all(got <= hi)
    print("worst relative difference", worst)
    eng.close()


def test_truncated_scan_properties_at_bench_size():
    """n = 10 000 (cfg3 size), in-kernel uniforms: the draw stays inside the limits, is reproducible, differs between
    chains, and one scan from a feasible start moves every coordinate."""
    n, C = 10000, 64
    eng = make_engine(C, seed=5)
    P_diag = np.full(n, 2.0)
    P_diag[0] = P_diag[-1] = 1.0
    P_diag[0] += 1e-3
    terms = [{"diag": eng.to_device(P_diag), "off": eng.full((n - 1,), -1.0), "scale": eng.full((C,), 50.0)},
             {"rhs": eng.to_device(np.sin(np.arange(n) / 300.0) + 1.0), "scale": eng.full((C,), 2.0)}]
    lower, upper = eng.full((n,), 0.2), eng.full((n,), 2.5)
    x0 = eng.full((C, n), 1.0)
    x1 = eng.tridiag_gibbs_truncated(n, terms, x0.clone(), lower=lower, upper=upper, draw_index=3)
    x2 = eng.tridiag_gibbs_truncated(n, terms, x0.clone(), lower=lower, upper=upper, draw_index=3)
    eng.check_status()
    a = x1.cpu().numpy()
    assert np.array_equal(a, x2.cpu().numpy())
    assert a.min() >= 0.2 and a.max() <= 2.5 and np.all(a != 1.0)
    assert np.abs(np.corrcoef(a[0], a[1])[0, 1]) < 0.9 and not np.array_equal(a[0], a[1])
    eng.close()


def test_normal_normal_with_truncated_prior_replays_reference(golden):
    """NormalNormal('b') whose prior has domain_response_lower = 1.5 inside the example-4 model (sparse route), plus
    both NormalGamma updates and log_post, 8 sweeps with the reference's uniforms and gammas."""
    import torch

    from openmcmc_amd.distribution.distribution import Gamma
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix
    from openmcmc_amd.sampler.sampler import NormalGamma, NormalNormal

    G = golden("truncated_conditional")
    n, S = int(G["mc_n"]), int(G["mc_sweeps"])
    P = sparse.diags((G["mc_P_off"], G["mc_P_diag"], G["mc_P_off"]), offsets=[-1, 0, 1], format="csc")
    mdl = Model([
        Normal("y", mean=LinearCombination(form={"b": "A"}), precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
        Normal("b", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda"),
               domain_response_lower=np.array(float(G["mc_lower"]))),
        Gamma("lambda", shape="a_lam", rate="b_lam"), Gamma("tau", shape="a_tau", rate="b_tau")])
    st = {"y": G["mc_y"].copy(), "b": G["mc_b0"].copy(), "mu": np.zeros(n), "lambda": 100, "P_lambda": P, "a_lam": 10, "b_lam": 1,
          "tau": 1, "P_tau": sparse.csc_matrix(np.eye(n)), "a_tau": 1, "b_tau": 1, "A": sparse.identity(n, format="csc")}
    C = 2
    dev = torch.device("cuda", 0)
    samplers = [NormalNormal("b", mdl), NormalGamma("lambda", mdl), NormalGamma("tau", mdl)]
    samplers[0].inject = lambda s, it: torch.as_tensor(np.tile(G["mc_u"][it], (C, 1)), device=dev)
    samplers[1].inject = lambda s, it: torch.full((C,), float(G["mc_g"][it, 0]), dtype=torch.float64, device=dev)
    samplers[2].inject = lambda s, it: torch.full((C,), float(G["mc_g"][it, 1]), dtype=torch.float64, device=dev)
    M = MCMC(st, samplers, model=mdl, n_burn=0, n_iter=S, n_chains=C)
    assert M._fused is None  # the truncated conditional is not part of the fused sweep
    M.run_mcmc()
    got = M.collect()
    for c in range(C):
        for key in ("b", "lambda", "tau", "log_post"):
            ref = G["mc_store_" + key]
            err = np.max(np.abs(got[key][c] - ref) / np.maximum(1.0, np.abs(ref)))
            assert err < RTOL, (key, err)
    assert got["b"].min() >= float(G["mc_lower"])


def test_log_p_is_minus_inf_outside_the_domain():
    """Normal.check_domain_response (location_scale.py:169-188): a chain whose response leaves [lower, upper] gets
    log_p = -inf, the others the un-normalised Gaussian density."""
    import torch

    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.model import Model
    from oracle import gmrf_ref

    n, C = 6, 4
    eng = make_engine(C)
    Q = sparse.diags(([-1.0] * (n - 1), [2.5] * n, [-1.0] * (n - 1)), offsets=[-1, 0, 1], format="csc")
    d = Normal("x", mean="mu", precision="Q", domain_response_lower=np.array(-1.0), domain_response_upper=np.array(2.0))
    rng = np.random.default_rng(0)
    x = rng.uniform(-0.9, 1.9, size=(C, n))
    x[1, 3] = 2.5   # above the upper limit
    x[2, 0] = -1.5  # below the lower limit
    state = {"x": ChainArray(eng.to_device(x)), "mu": np.zeros((n, 1)), "Q": Q}
    lp = Model([d]).log_p(state, engine=eng).cpu().numpy()
    eng.check_status()
    assert np.isneginf(lp[1]) and np.isneginf(lp[2])
    for c in (0, 3):
        ref = gmrf_ref.gauss_logpdf(x[c].reshape(n, 1), np.zeros((n, 1)), Q)
        assert abs(lp[c] - ref) < 1e-10 * max(1.0, abs(ref))
    eng.close()


def test_band_truncated_scan_matches_reference(golden):
    """One scan of single-site truncated updates under banded precisions wider than tridiagonal (RW2, bandwidth 3 and 5):
    gmrf.gibbs_canonical_truncated_normal run by the reference (tests/golden/band_truncated.npz), its uniforms injected."""
    from openmcmc_amd.engine import Engine

    G = golden("band_truncated")
    for ci in range(int(G["n_cases"])):
        k = f"c{ci}_"
        n, w, Q = int(G[k + "n"]), int(G[k + "w"]), G[k + "Q"]
        C = 3
        eng = Engine(C)
        band = np.zeros((w + 1, n))
        for d in range(w + 1):
            band[d, : n - d] = np.diag(Q, -d)
        x = eng.to_device(np.tile(G[k + "x0"], (C, 1)))
        lower, upper = float(G[k + "lower"]), float(G[k + "upper"])
        lo = None if np.isneginf(lower) else eng.full((n,), lower)
        hi = None if np.isposinf(upper) else eng.full((n,), upper)
        terms = [{"band": eng.to_device(band), "rhs": eng.to_device(G[k + "b"]), "scale": eng.full((C,), 1.0)}]
        eng.band_gibbs_truncated(n, terms, x, lower=lo, upper=hi, u=eng.to_device(np.tile(G[k + "u"], (C, 1))))
        eng.check_status()
        got = x.cpu().numpy()
        for c in range(C):
            assert np.max(np.abs(got[c] - G[k + "x"])) < 1e-9 * max(1.0, np.max(np.abs(G[k + "x"]))), (ci, c)
        eng.close()


@pytest.mark.parametrize("n,C,terms_n", [(150, 70, 2), (64, 3, 1), (1, 5, 2), (333, 65, 3)])
def test_tridiagonal_scan_blocks_per_chain_rhs_and_limits_per_site(n, C, terms_n):
    """The scan kernel's block structure (64 sites at a time through LDS tiles, groups of 8 sites, 64 chains per wave)
    at sizes that are not multiples of anything, with a per-chain right-hand side, 1 to 3 terms, limits that differ from
    site to site and bite at some sites (slow route) and not at others (far-limits route): against the CPU restatement of
    gmrf.gibbs_canonical_truncated_normal (oracle/gmrf_ref.py, pinned to the reference by tests/test_oracle_golden.py)
    with the same uniforms."""
    from oracle import gmrf_ref

    rng = np.random.default_rng(n * 31 + C)
    eng = make_engine(C, seed=9)
    d1 = 2.0 + rng.random(n)
    o1 = -0.8 * rng.random(max(n - 1, 0))
    lam = 1.0 + 3.0 * rng.random(C)
    tau = 0.5 + rng.random(C)
    y = rng.standard_normal(n)
    d3 = 0.3 * rng.random(n)
    rc = 0.5 * rng.standard_normal((C, n))
    lower = np.where(rng.random(n) < 0.5, -0.2, -30.0)
    upper = np.where(rng.random(n) < 0.3, 0.4, 40.0)
    x0 = np.clip(rng.standard_normal((C, n)), lower + 0.01, upper - 0.01)
    u = rng.random((C, n))
    terms = [{"diag": eng.to_device(d1), "off": eng.to_device(o1) if n > 1 else None, "scale": eng.to_device(lam)}]
    if terms_n >= 2:
        terms.append({"rhs": eng.to_device(y), "scale": eng.to_device(tau)})
    if terms_n >= 3:
        terms.append({"diag": eng.to_device(d3)})
    x = eng.to_device(x0)
    eng.tridiag_gibbs_truncated(n, terms, x, lower=eng.to_device(lower), upper=eng.to_device(upper), u=eng.to_device(u),
                                rhs_chain=eng.to_device(rc))
    eng.check_status()
    got = x.cpu().numpy()
    worst = 0.0
    for c in range(0, C, max(1, C // 7)):
        diag = lam[c] * d1 + (tau[c] if terms_n >= 2 else 0.0) + (d3 if terms_n >= 3 else 0.0)
        Q = np.diag(diag)
        if n > 1:
            Q += np.diag(lam[c] * o1, 1) + np.diag(lam[c] * o1, -1)
        b = rc[c] + (tau[c] * y if terms_n >= 2 else 0.0)
        ref = gmrf_ref.gibbs_truncated_scan(b, Q, x0[c], lower, upper, u[c]).ravel()
        worst = max(worst, np.max(np.abs(got[c] - ref) / np.maximum(1e-3, np.abs(ref))))
    assert worst < RTOL, worst
    assert np.all(got >= lower) and np.all(got <= upper)
    eng.close()
