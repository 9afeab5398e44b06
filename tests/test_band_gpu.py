"""Banded precisions of any bandwidth on the GPU: omc_band_sample_canonical /
omc_band_quadform against the CPU oracle (natural-order sparse factorisation, as the reference) with injected
draws, on RW2 precisions and random symmetric positive definite band matrices up to bandwidth 128."""

import numpy as np
import pytest
from scipy import sparse

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def make_engine(C, **kw):
    from openmcmc_amd.engine import Engine

    return Engine(C, **kw)


def band_of(M, w):
    """(w+1, n) array, row d = d-th sub-diagonal padded with zeros (the ABI's layout)."""
    M = M.toarray() if sparse.issparse(M) else np.asarray(M)
    n = M.shape[0]
    out = np.zeros((w + 1, n))
    for d in range(w + 1):
        out[d, : n - d] = np.diag(M, -d)
    return out


def rw2_precision(n):
    """Second-order random walk: D2' D2 with D2 the second-difference operator (pentadiagonal), plus a small ridge."""
    if n < 3:
        return sparse.identity(n, format="csc") * 1.0
    D = sparse.diags([np.ones(n - 2), -2 * np.ones(n - 2), np.ones(n - 2)], offsets=[0, 1, 2], shape=(n - 2, n))
    return (D.T @ D + 1e-3 * sparse.identity(n)).tocsc()


def random_band_spd(n, w, rng):
    A = np.zeros((n, n))
    for d in range(1, min(w, n - 1) + 1):
        v = rng.standard_normal(n - d) * 0.5
        A += np.diag(v, -d) + np.diag(v, d)
    A += np.diag(np.abs(A).sum(axis=1) + 0.5 + rng.random(n))
    return sparse.csc_matrix(A)


def check_case(eng, M, w, rng, C, per_chain_rhs=True):
    """per_chain_rhs=True exercises the workgroup-per-chain kernel (the only one that takes a per-chain right-hand
    side); False lets narrow bands (w <= 8) take the lane-per-chain kernel."""
    from oracle import gmrf_ref

    n = M.shape[0]
    lam, tau = rng.random(C) * 3 + 0.5, rng.random(C) * 2 + 0.2
    m = rng.standard_normal(n)
    y = rng.standard_normal(n)
    z = rng.standard_normal((C, n))
    extra = rng.standard_normal((C, n)) * 0.3 if per_chain_rhs else np.zeros((C, n))
    terms = [{"band": eng.to_device(band_of(M, w)), "rhs": eng.to_device(M @ m), "scale": eng.to_device(lam)},
             {"rhs": eng.to_device(y), "scale": eng.to_device(tau)}]
    x, mu, ld = eng.empty(C, n), eng.empty(C, n), eng.empty(C)
    eng.band_sample_canonical(n, terms, x, z=eng.to_device(z), rhs_chain=eng.to_device(extra) if per_chain_rhs else None,
                              mean_out=mu, logdet_out=ld)
    eng.check_status()
    xg, mg, lg = x.cpu().numpy(), mu.cpu().numpy(), ld.cpu().numpy()
    worst = 0.0
    for c in range(C):
        Q = (lam[c] * M + tau[c] * sparse.identity(n, format="csc")).tocsc()
        b = (lam[c] * (M @ m) + tau[c] * y + extra[c]).reshape(n, 1)
        xo, mo, L = gmrf_ref.draw_canonical(b, Q, z[c].reshape(n, 1))
        logdet = 2 * np.sum(np.log(L.diagonal()))
        scale = max(1.0, np.max(np.abs(xo)))
        worst = max(worst, np.max(np.abs(xg[c] - xo.ravel())) / scale, np.max(np.abs(mg[c] - mo.ravel())) / scale,
                    abs(lg[c] - logdet) / max(1.0, abs(logdet)))
    # quadratic form
    quad = eng.empty(C)
    eng.band_quadform(n, terms[0]["band"], x, quad, center=eng.to_device(m))
    r = xg - m[None, :]
    exp = np.einsum("ci,ci->c", r, (M @ r.T).T)
    worst = max(worst, np.max(np.abs(quad.cpu().numpy() - exp) / np.maximum(1.0, np.abs(exp))))
    return worst


@pytest.mark.parametrize("per_chain_rhs", [True, False], ids=["workgroup", "lane"])
@pytest.mark.parametrize("n", [1, 2, 3, 5, 40, 301])
def test_rw2_precision(n, per_chain_rhs):
    rng = np.random.default_rng(n)
    C = 5
    eng = make_engine(C)
    worst = check_case(eng, rw2_precision(n), 2 if n >= 3 else 0, rng, C, per_chain_rhs)
    assert worst < 1e-8, worst  # RW2 + ridge is ill-conditioned (cond ~ n^4): compared at its conditioning
    eng.close()


@pytest.mark.parametrize("w,n", [(0, 7), (1, 9), (3, 3), (3, 50), (7, 8), (7, 130), (20, 21), (20, 400), (64, 200),
                                  (100, 350), (128, 129), (128, 600)])
def test_random_band_matrices(w, n):
    rng = np.random.default_rng(100 * w + n)
    C = 4
    eng = make_engine(C)
    worst = check_case(eng, random_band_spd(n, w, rng), w, rng, C)
    assert worst < RTOL, worst
    eng.close()


@pytest.mark.parametrize("w,n,C", [(1, 9, 3), (2, 200, 70), (3, 3, 64), (4, 65, 65), (5, 129, 130), (6, 64, 7), (7, 130, 4),
                                    (8, 300, 100), (8, 9, 1)])
def test_lane_per_chain_kernel(w, n, C):
    """Narrow bands without a per-chain right-hand side take k_band_lane (window in registers, chains across the lanes):
    every bandwidth 1..8, chain counts around the wave size, lengths around the 64-column output tile."""
    rng = np.random.default_rng(1000 * w + n + C)
    eng = make_engine(C)
    worst = check_case(eng, random_band_spd(n, w, rng), w, rng, C, per_chain_rhs=False)
    assert worst < RTOL, worst
    # the two kernels agree with each other as well
    M = random_band_spd(n, w, rng)
    band = eng.to_device(band_of(M, w))
    z = eng.to_device(rng.standard_normal((C, n)))
    a, b = eng.empty(C, n), eng.empty(C, n)
    eng.band_sample_canonical(n, [{"band": band}], a, z=z)
    eng.set_option("band_algo", 2)
    eng.band_sample_canonical(n, [{"band": band}], b, z=z)
    eng.set_option("band_algo", 0)
    eng.check_status()
    assert np.max(np.abs(a.cpu().numpy() - b.cpu().numpy())) < 1e-11 * max(1.0, np.abs(b.cpu().numpy()).max())
    eng.close()


def test_band_failure_latch_and_in_kernel_draws():
    n, w, C = 60, 3, 3
    rng = np.random.default_rng(3)
    M = random_band_spd(n, w, rng)
    eng = make_engine(C, seed=9)
    band = eng.to_device(band_of(M, w))
    x = eng.empty(C, n)
    scale = eng.to_device(np.array([1.0, -1.0, 2.0]))  # chain 1: negative definite
    eng.band_sample_canonical(n, [{"band": band, "scale": scale}], x)
    with pytest.raises(np.linalg.LinAlgError, match="chain 1"):
        eng.check_status()
    eng2 = make_engine(C, seed=9)
    a, b = eng2.empty(C, n), eng2.empty(C, n)
    eng2.band_sample_canonical(n, [{"band": band}], a, draw_index=4)
    eng2.band_sample_canonical(n, [{"band": band}], b, draw_index=4)
    eng2.check_status()
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy()) and not np.array_equal(a[0].cpu().numpy(), a[1].cpu().numpy())
    eng.close(), eng2.close()


def test_rw2_smoother_replays_reference(golden):
    """The example-4 model with a second-order random-walk prior (pentadiagonal precision) through MCMC.run_mcmc:
    NormalNormal on the band route, NormalGamma with the band quadratic form, log_post with the band log-determinant;
    the reference's draws injected (tests/golden/band_chain.npz)."""
    import torch

    from openmcmc_amd.distribution.distribution import Gamma
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix
    from openmcmc_amd.sampler.sampler import NormalGamma, NormalNormal

    G = golden("band_chain")
    n, n_burn, n_iter = int(G["n"]), int(G["n_burn"]), int(G["n_iter"])
    mdl = Model([
        Normal("y", mean=LinearCombination(form={"b": "A"}), precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
        Normal("b", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
        Gamma("lambda", shape="a_lam", rate="b_lam"), Gamma("tau", shape="a_tau", rate="b_tau")])
    st = {"y": G["y"].copy(), "b": G["y"].copy(), "mu": np.full(n, float(G["mu"])), "lambda": 50,
          "P_lambda": sparse.csc_matrix(G["P"]), "a_lam": 10, "b_lam": 1, "tau": 1, "P_tau": sparse.csc_matrix(np.eye(n)),
          "a_tau": 1, "b_tau": 1, "A": sparse.identity(n, format="csc")}
    C = 2
    dev = torch.device("cuda", 0)
    samplers = [NormalNormal("b", mdl), NormalGamma("lambda", mdl), NormalGamma("tau", mdl)]
    samplers[0].inject = lambda s, it: torch.as_tensor(np.tile(G["z"][it], (C, 1)), device=dev)
    samplers[1].inject = lambda s, it: torch.full((C,), float(G["g"][it, 0]), dtype=torch.float64, device=dev)
    samplers[2].inject = lambda s, it: torch.full((C,), float(G["g"][it, 1]), dtype=torch.float64, device=dev)
    M = MCMC(st, samplers, model=mdl, n_burn=n_burn, n_iter=n_iter, n_chains=C)
    assert samplers[0].plan(M.state)["kind"] == "band"
    M.run_mcmc()
    got = M.collect()
    for c in range(C):
        for key in ("b", "lambda", "tau", "log_post"):
            ref = G["store_" + key]
            err = np.max(np.abs(got[key][c] - ref) / np.maximum(1.0, np.abs(ref)))
            assert err < 1e-9, (key, err)  # D2'D2 + 1e-3 I at n = 45 has condition number ~ 4e7


@pytest.mark.parametrize("tag", ["h", "o"])
def test_rw2_per_chain_rhs_pieces_replay_reference(golden, tag):
    """Band route with per-chain right-hand-side pieces (tests/golden/band_hier.npz): "h" a sampled prior mean, a sampled
    response under a scaled pentadiagonal precision and a residual with both sides sampled; "o" a smoother next to a regression
    block, each conditional seeing the other as an offset (sampler.py:181-192)."""
    from openmcmc_amd.distribution.distribution import Gamma
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix
    from openmcmc_amd.sampler.sampler import NormalGamma, NormalNormal

    G = golden("band_hier")
    n, k = int(G["n"]), tag + "_"
    if tag == "h":
        mdl = Model([
            Normal("y", mean="b", precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
            Normal("b", mean="m", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
            Normal("m", mean="m0", precision="P_m"),
            Gamma("lambda", shape="a_lam", rate="b_lam"), Gamma("tau", shape="a_tau", rate="b_tau")])
        normals, keys, cuts = [NormalNormal("b", mdl), NormalNormal("m", mdl)], ("b", "m", "lambda", "tau", "log_post"), [0, n, 2 * n]
    else:
        mdl = Model([
            Normal("y", mean=LinearCombination(form={"b": "A", "beta": "X"}), precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
            Normal("b", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
            Normal("beta", mean="mu_beta", precision="P_beta"),
            Gamma("lambda", shape="a_lam", rate="b_lam"), Gamma("tau", shape="a_tau", rate="b_tau")])
        normals, keys, cuts = [NormalNormal("b", mdl), NormalNormal("beta", mdl)], ("b", "beta", "lambda", "tau", "log_post"), [0, n, n + 3]
    gammas = [NormalGamma("lambda", mdl), NormalGamma("tau", mdl)]
    y = G["y"]
    state = {"y": y.copy(), "b": y.copy(), "m": np.full(n, 1.0), "m0": np.zeros(n), "P_m": sparse.csc_matrix(0.5 * np.eye(n)),
             "mu": np.zeros(n), "lambda": 20, "P_lambda": sparse.csc_matrix(G["P"]), "a_lam": 10, "b_lam": 1, "tau": 1,
             "P_tau": sparse.csc_matrix(np.eye(n)), "a_tau": 1, "b_tau": 1, "A": sparse.identity(n, format="csc"), "X": G["X"],
             "beta": np.zeros(3), "mu_beta": np.zeros(3), "P_beta": sparse.csc_matrix(np.diag([0.1, 0.2, 0.3]))}
    C = 3
    M = MCMC(state, normals + gammas, model=mdl, n_burn=int(G["n_burn"]), n_iter=int(G["n_iter"]), n_chains=C)
    eng = M.engine
    assert normals[0].plan(M.state)["kind"] == "band"
    for i, smp in enumerate(normals):
        smp.inject = lambda s_, t, i=i: eng.to_device(np.tile(G[k + "z"][t, cuts[i]:cuts[i + 1]], (C, 1)))
    for i, smp in enumerate(gammas):
        smp.inject = lambda s_, t, i=i: eng.full((C,), G[k + "g"][t, i])
    M.run_mcmc()
    got = M.collect()
    for c in range(C):
        for key in keys:
            ref = G[k + "store_" + key]
            err = np.max(np.abs(got[key][c] - ref) / np.maximum(1.0, np.abs(ref)))
            assert err < 1e-9, (key, c, err)


@pytest.mark.parametrize("tag", ["d", "w"])
def test_replicated_response_under_dense_and_band_precision(golden, tag):
    """n_rep columns of y under a response precision that is not diagonal (sampler.py:165-167,187-188; gmrf.py:346-348):
    dense (n = 8) and pentadiagonal (n = 14, band route); NormalNormal(x) and the stored log_post replay the reference
    (tests/golden/replicated_dense.npz)."""
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.sampler.sampler import NormalNormal

    G = golden("replicated_dense")
    k = tag + "_"
    n, C = int(G[k + "n"]), 2
    mdl = Model([Normal("y", mean="x", precision="Q_y"), Normal("x", mean="mu", precision="P_x")])
    state = {"y": G[k + "y"].copy(), "x": np.zeros(n), "mu": np.full(n, 0.3), "Q_y": sparse.csc_matrix(G[k + "Q_y"]),
             "P_x": sparse.csc_matrix(G[k + "P_x"])}
    smp = NormalNormal("x", mdl)
    M = MCMC(state, [smp], model=mdl, n_burn=0, n_iter=int(G[k + "n_iter"]), n_chains=C)
    eng = M.engine
    assert smp.plan(M.state)["kind"] == ("dense" if tag == "d" else "band")
    smp.inject = lambda s_, t: eng.to_device(np.tile(G[k + "z"][t], (C, 1)))
    M.run_mcmc()
    got = M.collect()
    for c in range(C):
        for key in ("x", "log_post"):
            ref = G[k + "store_" + key]
            assert np.max(np.abs(got[key][c] - ref) / np.maximum(1.0, np.abs(ref))) < 1e-10, (key, c)


@pytest.mark.parametrize("n,w", [(1, 0), (5, 0), (7, 2), (64, 3), (1000, 5), (257, 128)])
def test_band_matvec_chain_against_numpy(n, w):
    """omc_band_matvec_chain: out[c] (+)= scale[c] * M v_c for a shared symmetric band matrix in lower band storage, identity
    when no band is given; with and without per-chain scale, overwrite and accumulate."""
    from openmcmc_amd.engine import Engine

    w = min(w, n - 1)
    rng = np.random.default_rng(n + w)
    C = 5
    M = np.zeros((n, n))
    band = np.zeros((w + 1, n))
    for d in range(w + 1):
        v = rng.standard_normal(n - d)
        band[d, : n - d] = v
        M += np.diag(v, -d) + (np.diag(v, d) if d else 0)
    V, sc, base = rng.standard_normal((C, n)), rng.random(C) + 0.5, rng.standard_normal((C, n))
    eng = Engine(C)
    dB, dV = eng.to_device(band), eng.to_device(V)
    got = eng.band_matvec_chain(n, dB, dV).cpu().numpy()
    assert np.allclose(got, V @ M, rtol=1e-13, atol=1e-13)
    out = eng.to_device(base.copy())
    eng.band_matvec_chain(n, dB, dV, scale=eng.to_device(sc), out=out, accumulate=True)
    assert np.allclose(out.cpu().numpy(), base + sc[:, None] * (V @ M), rtol=1e-13, atol=1e-13)
    ident = eng.band_matvec_chain(n, None, dV, scale=eng.to_device(sc)).cpu().numpy()
    assert np.allclose(ident, sc[:, None] * V, rtol=1e-15, atol=0)
    eng.close()


def test_long_tridiagonal_chains_take_the_band_route():
    """Beyond the 16 384 nodes one workgroup takes, NormalNormal plans the band route (w = 1: the segmented lane-per-chain
    kernels) instead of the tridiagonal route's one-lane-per-chain fallback.  Same draws: three sweeps of the example-4 model
    at n = 17 000 through MCMC.run_mcmc with injected draws, once on each route, agree to rounding (b, lambda, tau, log_post)."""
    from openmcmc_amd.distribution.distribution import Gamma
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.mcmc import MCMC
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix
    from openmcmc_amd.sampler.sampler import NormalGamma, NormalNormal

    n, C, sweeps = 17000, 3, 3
    rng = np.random.default_rng(5)
    t = np.arange(n) * 60.0 / 10000
    y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + 0.3 * rng.standard_normal(n)
    D = sparse.diags([-np.ones(n - 1), np.ones(n - 1)], offsets=[0, 1], shape=(n - 1, n))
    P = (D.T @ D + 1e-3 * sparse.identity(n)).tocsc()
    z, g = rng.standard_normal((sweeps, n)), 5.0 + rng.random((sweeps, 2))
    out = {}
    saved = NormalNormal.TRIDIAG_WG_MAX
    try:
        for route, limit in (("band", saved), ("tridiag", 10**9)):
            NormalNormal.TRIDIAG_WG_MAX = limit
            mdl = Model([Normal("y", mean=LinearCombination(form={"b": "A"}), precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
                         Normal("b", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
                         Gamma("lambda", shape="a_lam", rate="b_lam"), Gamma("tau", shape="a_tau", rate="b_tau")])
            state = {"y": y, "b": y.copy(), "mu": np.full(n, 0.3), "lambda": 50.0, "P_lambda": P, "a_lam": 10.0, "b_lam": 1.0,
                     "tau": 1.0, "P_tau": sparse.identity(n, format="csc"), "a_tau": 1.0, "b_tau": 1.0,
                     "A": sparse.identity(n, format="csc")}
            samplers = [NormalNormal("b", mdl), NormalGamma("lambda", mdl), NormalGamma("tau", mdl)]
            M = MCMC(state, samplers, model=mdl, n_burn=0, n_iter=sweeps, n_chains=C)
            eng = M.engine
            if route == "tridiag":  # the one-lane-per-chain kernel (since round 3 the C entry points reroute long chains as well)
                eng.set_option("tridiag_algo", 1)
            samplers[0].inject = lambda s_, it: eng.to_device(np.tile(z[it], (C, 1)))
            samplers[1].inject = lambda s_, it: eng.full((C,), g[it, 0])
            samplers[2].inject = lambda s_, it: eng.full((C,), g[it, 1])
            assert samplers[0].plan(M.state)["kind"] == route
            M.run_mcmc()
            out[route] = M.collect()
    finally:
        NormalNormal.TRIDIAG_WG_MAX = saved
    for key in ("b", "lambda", "tau", "log_post"):
        a, b = out["band"][key], out["tridiag"][key]
        assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) < 1e-9, key


@pytest.mark.parametrize("w,n", [(1, 9), (3, 50), (7, 130), (9, 10), (9, 200), (12, 333), (15, 16), (16, 17), (17, 100), (20, 21), (20, 400), (31, 95), (64, 200),
                                  (100, 350), (112, 113), (120, 250), (124, 300), (128, 129), (128, 600)])
def test_blocked_wide_band_kernel(w, n):
    """The blocked workgroup-per-chain kernel (omc_bandwide.hip: 16 columns per step -- 8 where the window would not fit the LDS
    --, diagonal block in registers, window update on the matrix cores), forced for every bandwidth: against the oracle like the
    column-at-a-time kernel, on lengths that end inside a block, with and without a per-chain right-hand side, and against that
    kernel itself."""
    rng = np.random.default_rng(100 * w + n + 1)
    C = 4
    eng = make_engine(C)
    eng.set_option("band_algo", 3)
    for per_chain in (True, False):
        worst = check_case(eng, random_band_spd(n, w, rng), w, rng, C, per_chain_rhs=per_chain)
        assert worst < RTOL, (worst, per_chain)
    M = random_band_spd(n, w, rng)
    band = eng.to_device(band_of(M, w))
    z = eng.to_device(rng.standard_normal((C, n)))
    extra = eng.to_device(rng.standard_normal((C, n)))
    a, b, ma, mb, la, lb = eng.empty(C, n), eng.empty(C, n), eng.empty(C, n), eng.empty(C, n), eng.empty(C), eng.empty(C)
    eng.band_sample_canonical(n, [{"band": band}], a, z=z, rhs_chain=extra, mean_out=ma, logdet_out=la)
    eng.set_option("band_algo", 2)
    eng.band_sample_canonical(n, [{"band": band}], b, z=z, rhs_chain=extra, mean_out=mb, logdet_out=lb)
    eng.check_status()
    scale = max(1.0, np.abs(b.cpu().numpy()).max())
    assert np.max(np.abs(a.cpu().numpy() - b.cpu().numpy())) < 1e-11 * scale
    assert np.max(np.abs(ma.cpu().numpy() - mb.cpu().numpy())) < 1e-11 * scale
    assert np.max(np.abs(la.cpu().numpy() - lb.cpu().numpy())) < 1e-12 * max(1.0, np.abs(lb.cpu().numpy()).max())
    eng.close()


@pytest.mark.parametrize("w,n,algo", [(20, 157, 3), (100, 260, 0), (124, 300, 0), (5, 90, 3)])
def test_blocked_wide_band_kernel_four_terms(w, n, algo):
    """Three and four terms (the kernel is compiled once for up to two terms and once for OMC_MAX_TERMS): two band matrices of
    different bandwidths and two identity terms with right-hand sides of their own, per-chain scales on all of them."""
    from oracle import gmrf_ref

    rng = np.random.default_rng(w * 1000 + n)
    C = 3
    eng = make_engine(C)
    eng.set_option("band_algo", algo)
    w2 = max(1, w // 3)
    M1, M2 = random_band_spd(n, w, rng), random_band_spd(n, w2, rng)
    m1, y3, y4 = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    for n_terms in (3, 4):
        sc = rng.random((4, C)) * 2 + 0.3
        terms = [{"band": eng.to_device(band_of(M1, w)), "rhs": eng.to_device(M1 @ m1), "scale": eng.to_device(sc[0])},
                 {"band": eng.to_device(band_of(M2, w2)), "scale": eng.to_device(sc[1])},
                 {"rhs": eng.to_device(y3), "scale": eng.to_device(sc[2])},
                 {"rhs": eng.to_device(y4), "scale": eng.to_device(sc[3])}][:n_terms]
        z, extra = rng.standard_normal((C, n)), rng.standard_normal((C, n))
        x, mu, ld = eng.empty(C, n), eng.empty(C, n), eng.empty(C)
        eng.band_sample_canonical(n, terms, x, z=eng.to_device(z), rhs_chain=eng.to_device(extra), mean_out=mu, logdet_out=ld)
        eng.check_status()
        for c in range(C):
            Q = sc[0, c] * M1 + sc[1, c] * M2 + sc[2, c] * sparse.identity(n)
            b = sc[0, c] * (M1 @ m1) + sc[2, c] * y3 + extra[c]
            if n_terms == 4:
                Q = Q + sc[3, c] * sparse.identity(n)
                b = b + sc[3, c] * y4
            xo, mo, L = gmrf_ref.draw_canonical(b.reshape(n, 1), sparse.csc_matrix(Q), z[c].reshape(n, 1))
            scale = max(1.0, np.max(np.abs(xo)))
            assert np.max(np.abs(x[c].cpu().numpy() - xo.ravel())) < RTOL * scale
            assert np.max(np.abs(mu[c].cpu().numpy() - mo.ravel())) < RTOL * scale
            logdet = 2 * np.sum(np.log(L.diagonal()))
            assert abs(float(ld[c]) - logdet) < RTOL * max(1.0, abs(logdet))
    eng.close()


@pytest.mark.parametrize("w,n,form", [(5, 130, 4), (12, 333, 4), (15, 48, 4), (9, 200, 8), (20, 400, 8), (40, 95, 8), (64, 200, 8), (16, 17, 8),
                                      (9, 200, 16), (20, 400, 16), (44, 95, 16), (33, 200, 16), (16, 17, 16), (12, 333, 512), (64, 200, 512),
                                      (1, 50, 8), (3, 200, 8), (2, 100, 16), (1, 90, 16), (1, 9, 512), (3, 50, 4), (1, 70, 4)])
def test_blocked_wide_band_kernel_forms_for_many_chains(w, n, form):
    """The forms of the blocked kernel the library picks when there are more chains than CUs -- four waves per chain at 128
    registers (bands narrower than a block), 8 columns per step at 128 registers with four or eight waves (bands up to ~64) -- and the one-workgroup-per-CU
    form, each forced: against the oracle, and against the column-at-a-time kernel."""
    rng = np.random.default_rng(7 * w + n + form)
    C = 5
    eng = make_engine(C)
    eng.set_option("band_algo", 3)
    eng.set_option("band_blocked_threads", form)
    for per_chain in (True, False):
        worst = check_case(eng, random_band_spd(n, w, rng), w, rng, C, per_chain_rhs=per_chain)
        assert worst < RTOL, (worst, per_chain)
    M = random_band_spd(n, w, rng)
    band = eng.to_device(band_of(M, w))
    z = eng.to_device(rng.standard_normal((C, n)))
    a, b, la, lb = eng.empty(C, n), eng.empty(C, n), eng.empty(C), eng.empty(C)
    eng.band_sample_canonical(n, [{"band": band}], a, z=z, logdet_out=la)
    eng.set_option("band_algo", 2)
    eng.band_sample_canonical(n, [{"band": band}], b, z=z, logdet_out=lb)
    eng.check_status()
    assert np.max(np.abs(a.cpu().numpy() - b.cpu().numpy())) < 1e-11 * max(1.0, np.abs(b.cpu().numpy()).max())
    assert np.max(np.abs(la.cpu().numpy() - lb.cpu().numpy())) < 1e-12 * max(1.0, np.abs(lb.cpu().numpy()).max())
    eng.close()


@pytest.mark.parametrize("n,w,C,form", [(700, 100, 6, 0), (3000, 8, 1100, 0), (3000, 8, 1100, 4), (2000, 12, 1500, 0), (3000, 32, 1100, 0),
                                        (3000, 32, 600, 8), (1500, 3, 900, 0)])
def test_blocked_wide_band_kernel_repeats_itself_bit_for_bit(n, w, C, form):
    """The waves of the blocked kernel take their shares of a block's window update from a counter (who takes which differs from
    run to run) and the backward pass sums through LDS: the orders of summation are fixed all the same, so a repeated call
    returns the same bits -- draw, mean and log det.  With more chains than the CUs hold workgroups at once, on every form of the
    kernel: workgroups then start on CUs others have left and run four to a CU, which is where a hand-over between waves that is
    not fenced shows (one did: the solved block was copied out of a ring another wave was already refilling, bands narrower than
    a block only)."""
    rng = np.random.default_rng(77)
    eng = make_engine(C)
    eng.set_option("band_algo", 3)
    eng.set_option("band_blocked_threads", form)
    M = random_band_spd(n, w, rng)
    terms = [{"band": eng.to_device(band_of(M, w)), "scale": eng.to_device(rng.random(C) + 0.5)},
             {"rhs": eng.to_device(rng.standard_normal(n)), "scale": eng.to_device(rng.random(C) + 0.5)}]
    z, extra = eng.to_device(rng.standard_normal((C, n))), eng.to_device(rng.standard_normal((C, n)))
    out = []
    for rep in range(4):
        x, mu, ld = eng.empty(C, n), eng.empty(C, n), eng.empty(C)
        eng.band_sample_canonical(n, terms, x, z=z, rhs_chain=extra, mean_out=mu, logdet_out=ld)
        eng.check_status()
        out.append((x.cpu().numpy(), mu.cpu().numpy(), ld.cpu().numpy()))
    for rep in range(1, 4):
        for a, b in zip(out[0], out[rep]):
            assert np.array_equal(a, b)
    eng.close()


def test_blocked_wide_band_kernel_latches_a_failed_chain_and_draws_in_kernel():
    n, w, C = 300, 40, 3
    rng = np.random.default_rng(5)
    M = random_band_spd(n, w, rng)
    eng = make_engine(C, seed=9)
    band = eng.to_device(band_of(M, w))
    x, y = eng.empty(C, n), eng.empty(C, n)
    eng.band_sample_canonical(n, [{"band": band}], x, draw_index=4)      # auto: the blocked kernel from w = 9
    eng.set_option("band_algo", 2)
    eng.band_sample_canonical(n, [{"band": band}], y, draw_index=4)      # same streams through the column-at-a-time kernel
    eng.check_status()
    assert np.max(np.abs(x.cpu().numpy() - y.cpu().numpy())) < 1e-11 * max(1.0, np.abs(y.cpu().numpy()).max())
    eng.set_option("band_algo", 0)
    scale = eng.to_device(np.array([1.0, -1.0, 2.0]))  # chain 1: negative definite
    eng.band_sample_canonical(n, [{"band": band, "scale": scale}], x)
    with pytest.raises(np.linalg.LinAlgError, match="chain 1"):
        eng.check_status()
    eng.close()
