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


@pytest.mark.parametrize("tag", ["scaled", "mixture"])
def test_manifold_mala_on_regression_coefficients(golden, tag):
    """ManifoldMALA where the Hessian is a per-chain combination of shared matrices: regression coefficients under a
    ScaledMatrix Gaussian prior or under a mixture prior (likelihood through the mean: grad_log_p branch ii).  40 steps
    of the reference replayed (tests/golden/mala_regression.npz): same accept decisions, states to 1e-9."""
    import torch
    from scipy import sparse

    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, MixtureParameterMatrix, MixtureParameterVector, ScaledMatrix
    from openmcmc_amd.sampler.metropolis_hastings import ManifoldMALA

    G = golden("mala_regression")
    X, w, y = G["X"], G["w"], G["y"]
    n, p = X.shape
    C = 3
    eng = make_engine(C)
    lik = Normal("y", mean=LinearCombination({"beta": "X"}), precision=ScaledMatrix("P_tau", "tau"))
    if tag == "scaled":
        prior = Normal("beta", mean="mu", precision=ScaledMatrix("P_lam", "lam"))
    else:
        prior = Normal("beta", mean=MixtureParameterVector("prior_mean", "alloc"), precision=MixtureParameterMatrix("prior_prec", "alloc"))
    mdl = Model([lik, prior])
    dev = eng.device
    full = lambda v: ChainArray(eng.full((C, 1, 1), float(v)))  # noqa: E731
    state = {"y": y.reshape(n, 1), "X": X, "beta": ChainArray(eng.to_device(np.tile(G["beta0"], (C, 1)))),
             "P_tau": sparse.diags(w, format="csc"), "tau": full(G["tau"]), "P_lam": G["P"], "lam": full(G["lam"]),
             "mu": np.full((p, 1), float(G["mu"])), "prior_mean": G["prior_mean"].reshape(3, 1),
             "prior_prec": G["prior_prec"].reshape(3, 1), "alloc": ChainArray(eng.to_device(np.tile(G["alloc"], (C, 1))))}
    smp = ManifoldMALA("beta", mdl, step=np.array(float(G["step"]))).bind(eng)
    smp.inject = lambda s, it: torch.as_tensor(np.tile(G[tag + "_z"][it], (C, 1)), device=dev)
    smp.inject_uniform = lambda s, it: torch.full((C,), float(G[tag + "_u"][it]), dtype=torch.float64, device=dev)
    for it in range(int(G["n_steps"])):
        before = smp.accept_rate.accept.clone()
        state = smp.sample(state)
        eng.check_status()
        got = state["beta"].numpy()[:, :, 0]
        ref = G[tag + "_x"][it]
        assert np.max(np.abs(got - ref[None, :])) < 1e-9 * max(1.0, np.abs(ref).max()), (tag, it)
        assert (smp.accept_rate.accept - before).cpu().numpy().tolist() == [int(G[tag + "_accept"][it])] * C, (tag, it)
    eng.close()


@pytest.mark.parametrize("d,C", [(137, 70), (500, 33), (64, 16)])
@pytest.mark.parametrize("kind", ["mala", "rw"])
def test_own_gemm_route_matches_rocblas_route(d, C, kind):
    """The products of the fused steps on the own small-tile fp64 MFMA GEMM (omc_gemm.hip; partial tiles in both
    directions, the fused two-pair launch, the vector epilogue with a non-zero mean, the triangular skip) against the same
    steps through rocBLAS: same injected draws, states equal to rounding, decisions identical."""
    import torch

    rng = np.random.default_rng(d + C)
    A = rng.standard_normal((d, 2 * d))
    Qh = np.linalg.inv(A @ A.T / (2 * d))
    Qh = (Qh + Qh.T) / 2
    mu = rng.standard_normal(d)
    x0 = mu + np.linalg.solve(np.linalg.cholesky(Qh).T, rng.standard_normal((d, C))).T
    step = 0.5 if kind == "mala" else 0.05
    zs, us = rng.standard_normal((6, C, d)), rng.random((6, C))
    out = {}
    for flag in (0, 1):
        eng = make_engine(C)
        eng.set_option("mh_use_rocblas", flag)
        Q, dmu = eng.to_device(Qh), eng.to_device(mu)
        L, sl = eng.dense_cholesky(Q, 1.0 / step**2 if kind == "mala" else 1.0)
        x = eng.to_device(x0)
        acc = torch.zeros(C, dtype=torch.int64, device="cuda")
        prop = torch.zeros(C, dtype=torch.int64, device="cuda")
        for i in range(6):
            z, u = eng.to_device(zs[i]), eng.to_device(us[i])
            if kind == "mala":
                eng.mala_step(Q, dmu, L, sl, step, x, z=z, u=u, accept_count=acc, proposal_count=prop)
            else:
                eng.rw_step(dmu, L, sl, step, x, z=z, u=u, accept_count=acc, proposal_count=prop)
        eng.check_status()
        out[flag] = (x.cpu().numpy(), acc.cpu().numpy())
        eng.close()
    assert relerr(out[0][0], out[1][0]) < 1e-11
    assert np.array_equal(out[0][1], out[1][1])
    assert 0 < out[0][1].sum() < 6 * C


def test_manifold_mala_with_finite_difference_hessian_replays_reference(golden):
    """ManifoldMALA on a vector with an element-wise Gamma prior under a regression likelihood: the prior has no analytic
    gradient, so gradient AND Hessian come from the base class's central differences (distribution.py:90-198) and the
    Hessian depends on the parameter -- the generic route (per-chain H_c, natural-order factor per chain).  Replay of the
    reference's 30 steps (tests/golden/mala_fd.npz) with its z and u injected.  Finite differences with h = 1e-4 amplify
    rounding by 1/h^2, in the reference as here, so states agree to ~1e-6, decisions exactly."""
    import torch

    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.distribution import Gamma
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.engine import Engine
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix
    from openmcmc_amd.sampler.metropolis_hastings import ManifoldMALA
    from scipy import sparse

    G = golden("mala_fd")
    C, d, n_obs = 3, 3, G["A"].shape[0]
    eng = Engine(C, seed=1)
    mdl = Model([
        Normal("y", mean=LinearCombination(form={"s": "A"}), precision=ScaledMatrix(matrix="P_y", scalar="tau")),
        Gamma("s", shape="a_s", rate="b_s"),
    ])
    state = {"y": G["y"].reshape(-1, 1), "A": G["A"], "s": ChainArray(eng.to_device(np.tile(G["s0"], (C, 1)))),
             "tau": ChainArray(eng.full((C, 1, 1), float(G["tau"]))), "P_y": sparse.csc_matrix(np.eye(n_obs)),
             "a_s": float(G["a_s"]), "b_s": float(G["b_s"])}
    smp = ManifoldMALA("s", mdl, step=np.array([float(G["step"])]))
    smp.bind(eng, 0, 1)
    smp.inject = lambda s_, t: eng.to_device(np.tile(G["z"][t], (C, 1)))
    smp.inject_uniform = lambda s_, t: eng.full((C,), G["u"][t])
    smp.trace = {}
    flags = []
    for t in range(int(G["n_steps"])):
        before = smp.accept_rate.accept.clone()
        state = smp.sample(state)
        flags.append((smp.accept_rate.accept - before).cpu().numpy())
        step = smp.trace["steps"][-1]
        assert relerr(step["prop"][1].cpu().numpy(), G["prop"][t]) < 5e-6, t
        assert abs(step["lq_fwd"][1].item() - G["lq_fwd"][t]) < 1e-4 and abs(step["lq_rev"][1].item() - G["lq_rev"][t]) < 1e-4, t
        assert relerr(state["s"].data[2, :, 0].cpu().numpy(), G["x"][t]) < 5e-6, t
    eng.check_status()
    flags = np.array(flags)
    for c in range(C):
        assert np.array_equal(flags[:, c], G["accept"])
    eng.close()


@pytest.mark.parametrize("d", [1, 5, 32])
@pytest.mark.parametrize("reuse", [False, True])
def test_whitened_mala_step_replays_reference(golden, d, reuse):
    """omc_mala_step_white (the step in a = L'(x - mu), one triangular product per step) on the reference's trace
    (tests/golden/mala.npz): accept flags identical, states to 1e-9 -- with the whitened state carried from step to step
    and with it recomputed from x every step."""
    import torch

    G = golden("mala")
    k = f"d{d}_"
    C = 3
    eng = make_engine(C)
    Q = eng.to_device(G[k + "Q"])
    step = float(G[k + "mala_step"])
    x = eng.to_device(np.tile(G[k + "x0"], (C, 1)))
    acc = torch.zeros(C, dtype=torch.int64, device="cuda")
    prop = torch.zeros(C, dtype=torch.int64, device="cuda")
    L, sl = eng.dense_cholesky(Q, 1.0 / step**2)
    zs, us = G[k + "mala_z"], G[k + "mala_u"]
    flags = []
    for i in range(zs.shape[0]):
        before = acc.clone()
        z = eng.to_device(np.tile(zs[i], (C, 1)))
        u = eng.full((C,), us[i])
        eng.mala_step_white(None, L, sl, step, x, state_is_current=reuse and i > 0, z=z, u=u, accept_count=acc,
                            proposal_count=prop)
        flags.append((acc - before).cpu().numpy())
        assert relerr(x[2].cpu().numpy(), G[k + "mala_x"][i]) < 1e-9, i
    eng.check_status()
    flags = np.array(flags)
    for c in range(C):
        assert np.array_equal(flags[:, c], G[k + "mala_accept"])
    eng.close()


@pytest.mark.parametrize("d,C", [(137, 70), (500, 33), (64, 16)])
def test_whitened_mala_step_matches_the_products_route(d, C):
    """Same target (non-zero mean), same injected draws: omc_mala_step_white against omc_mala_step -- decisions identical,
    states equal to rounding; rejected chains keep their x bit for bit; an x written by someone else between two steps is
    picked up when the caller says so."""
    import torch

    rng = np.random.default_rng(d * 7 + C)
    A = rng.standard_normal((d, 2 * d))
    Qh = np.linalg.inv(A @ A.T / (2 * d))
    Qh = (Qh + Qh.T) / 2
    mu = rng.standard_normal(d)
    x0 = mu + np.linalg.solve(np.linalg.cholesky(Qh).T, rng.standard_normal((d, C))).T
    step = 0.5
    zs, us = rng.standard_normal((8, C, d)), rng.random((8, C))
    x_mid = mu + np.linalg.solve(np.linalg.cholesky(Qh).T, rng.standard_normal((d, C))).T  # written "by another sampler"
    out = {}
    for white in (0, 1):
        eng = make_engine(C)
        Q, dmu = eng.to_device(Qh), eng.to_device(mu)
        L, sl = eng.dense_cholesky(Q, 1.0 / step**2)
        x = eng.to_device(x0)
        acc = torch.zeros(C, dtype=torch.int64, device="cuda")
        prop = torch.zeros(C, dtype=torch.int64, device="cuda")
        kept = True
        for i in range(8):
            if i == 5:
                x.copy_(eng.to_device(x_mid))
            z, u = eng.to_device(zs[i]), eng.to_device(us[i])
            before, xb = acc.clone(), x.clone()
            if white:
                eng.mala_step_white(dmu, L, sl, step, x, state_is_current=i not in (0, 5), z=z, u=u, accept_count=acc,
                                    proposal_count=prop)
            else:
                eng.mala_step(Q, dmu, L, sl, step, x, z=z, u=u, accept_count=acc, proposal_count=prop)
            rejected = (acc - before) == 0
            kept = kept and bool(torch.equal(x[rejected], xb[rejected]))
        eng.check_status()
        out[white] = (x.cpu().numpy(), acc.cpu().numpy(), kept)
        eng.close()
    assert out[0][2] and out[1][2]
    assert np.array_equal(out[0][1], out[1][1])
    assert relerr(out[0][0], out[1][0]) < 1e-10
    assert 0 < out[1][1].sum() < 8 * C


@pytest.mark.parametrize("d,C", [(137, 70), (500, 33), (5, 3)])
def test_whitened_rw_step_matches_the_products_route(d, C):
    """omc_rw_step_white (L_Q'(x - mu) carried, one triangular product per step) against omc_rw_step on a target with a
    non-zero mean and the same injected draws: decisions identical, states BIT-identical (x' = x + step z is formed the
    same way in both), also across a write to x by someone else."""
    import torch

    rng = np.random.default_rng(d * 3 + C)
    A = rng.standard_normal((d, 2 * d))
    Qh = np.linalg.inv(A @ A.T / (2 * d))
    Qh = (Qh + Qh.T) / 2
    mu = rng.standard_normal(d)
    x0 = mu + np.linalg.solve(np.linalg.cholesky(Qh).T, rng.standard_normal((d, C))).T
    x_mid = mu + np.linalg.solve(np.linalg.cholesky(Qh).T, rng.standard_normal((d, C))).T
    step = 0.05
    zs, us = rng.standard_normal((8, C, d)), rng.random((8, C))
    out = {}
    for white in (0, 1):
        eng = make_engine(C)
        Q, dmu = eng.to_device(Qh), eng.to_device(mu)
        L, sl = eng.dense_cholesky(Q, 1.0)
        x = eng.to_device(x0)
        acc = torch.zeros(C, dtype=torch.int64, device="cuda")
        prop = torch.zeros(C, dtype=torch.int64, device="cuda")
        for i in range(8):
            if i == 5:
                x.copy_(eng.to_device(x_mid))
            z, u = eng.to_device(zs[i]), eng.to_device(us[i])
            if white:
                eng.rw_step_white(dmu, L, sl, step, x, state_is_current=i not in (0, 5), z=z, u=u, accept_count=acc,
                                  proposal_count=prop)
            else:
                eng.rw_step(dmu, L, sl, step, x, z=z, u=u, accept_count=acc, proposal_count=prop)
        eng.check_status()
        out[white] = (x.cpu().numpy(), acc.cpu().numpy())
        eng.close()
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][0], out[1][0])
    assert 0 < out[1][1].sum() < 8 * C


@pytest.mark.parametrize("case", ["r", "m"])
def test_manifold_mala_through_lognormal_gradients(golden, case):
    """LogNormal's analytic gradient and Hessian (location_scale.py:302-402): "r" the sampled vector is the response of a
    log-normal prior (Hessian depends on the vector: generic route, one H_c per chain), "m" the sampled coefficients enter
    the mean of a log-normal likelihood (constant Hessian: dense route).  30 reference steps each, z and u injected
    (tests/golden/mala_lognormal.npz): decisions identical, states to 1e-8."""
    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.location_scale import LogNormal, Normal
    from openmcmc_amd.engine import Engine
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix
    from openmcmc_amd.sampler.metropolis_hastings import ManifoldMALA
    from scipy import sparse

    G = golden("mala_lognormal")
    C, n_obs = 3, G["A"].shape[0]
    eng = Engine(C, seed=1)
    k = case + "_"
    if case == "r":
        mdl = Model([Normal("y", mean=LinearCombination(form={"s": "A"}), precision=ScaledMatrix(matrix="P_y", scalar="tau")),
                     LogNormal("s", mean="mu_s", precision="Q_s")])
        prm, step = "s", float(G["step"])
        state = {"y": G["y"].reshape(-1, 1), "A": G["A"], "s": ChainArray(eng.to_device(np.tile(G["s0"], (C, 1)))),
                 "tau": ChainArray(eng.full((C, 1, 1), float(G["tau"]))), "P_y": sparse.csc_matrix(np.eye(n_obs)),
                 "mu_s": G["mu_s"].reshape(-1, 1), "Q_s": G["Q_s"]}
    else:
        mdl = Model([LogNormal("w", mean=LinearCombination(form={"beta": "X"}), precision=ScaledMatrix(matrix="P_w", scalar="tau_w")),
                     Normal("beta", mean="mu_b", precision="P_b")])
        prm, step = "beta", float(G["step_b"])
        state = {"w": G["w"].reshape(-1, 1), "X": G["X"], "beta": ChainArray(eng.to_device(np.tile(G["beta0"], (C, 1)))),
                 "tau_w": ChainArray(eng.full((C, 1, 1), float(G["tau_w"]))), "P_w": sparse.csc_matrix(np.eye(n_obs)),
                 "mu_b": np.zeros((2, 1)), "P_b": sparse.csc_matrix(0.5 * np.eye(2))}
    smp = ManifoldMALA(prm, mdl, step=np.array([step]))
    smp.bind(eng, 0, 1)
    smp.inject = lambda s_, t: eng.to_device(np.tile(G[k + "z"][t], (C, 1)))
    smp.inject_uniform = lambda s_, t: eng.full((C,), G[k + "u"][t])
    flags = []
    for t in range(int(G["n_steps"])):
        before = smp.accept_rate.accept.clone()
        state = smp.sample(state)
        flags.append((smp.accept_rate.accept - before).cpu().numpy())
        assert relerr(state[prm].data[2, :, 0].cpu().numpy(), G[k + "x"][t]) < 1e-8, t
    eng.check_status()
    flags = np.array(flags)
    for c in range(C):
        assert np.array_equal(flags[:, c], G[k + "accept"])
    eng.close()


def test_whitened_mala_at_the_cfg4_size():
    """BASELINE configs[3] at its per-GPU size: d = 500, 512 chains, step 0.5, the whitened step with in-kernel draws and its
    cached whitened state carried: the acceptance rate of this target at stationarity (41 000 proposals here: 0.727 +- 0.003;
    the CPU oracle's single chain gives 0.72 +- 0.02 over 400 steps; SURVEY.md 3.4 quotes ~68 % from a short reference run)
    and the chains stay in the target, E|L_Q'(x - mu)|^2 = d, with a non-zero mean."""
    import torch

    d, C = 500, 512
    rng = np.random.default_rng(0)
    A = rng.standard_normal((d, 2 * d))
    Qh = np.linalg.inv(A @ A.T / (2 * d))
    Qh = (Qh + Qh.T) / 2
    mu = rng.standard_normal(d)
    eng = make_engine(C, seed=11)
    Q, dmu = eng.to_device(Qh), eng.to_device(mu)
    step = 0.5
    L, sl = eng.dense_cholesky(Q, 1.0 / step**2)
    x = eng.to_device(mu + np.linalg.solve(np.linalg.cholesky(Qh).T, rng.standard_normal((d, C))).T)  # draws from the target
    acc = torch.zeros(C, dtype=torch.int64, device="cuda")
    prop = torch.zeros(C, dtype=torch.int64, device="cuda")
    lp = eng.empty(C)
    n_steps = 80
    for it in range(n_steps):
        eng.mala_step_white(dmu, L, sl, step, x, state_is_current=it > 0, draw_index=it, accept_count=acc, proposal_count=prop,
                            log_p_out=lp)
    eng.check_status()
    assert np.array_equal(prop.cpu().numpy(), np.full(C, n_steps))
    rate = acc.sum().item() / prop.sum().item()
    assert 0.705 < rate < 0.75, rate
    r = x.cpu().numpy() - mu
    maha = np.einsum("ci,ij,cj->c", r, Qh, r)
    assert abs(maha.mean() / d - 1) < 0.015
    # the log density the step hands out is the target's at the state it left (gmrf.py:321-348)
    want = 0.5 * (np.linalg.slogdet(Qh)[1] - d * np.log(2 * np.pi) - maha)
    assert np.max(np.abs(lp.cpu().numpy() - want)) < 1e-8 * d
    eng.close()


@pytest.mark.parametrize("cls_name", ["ManifoldMALA", "RandomWalk"])
def test_cached_whitened_state_notices_library_writes(cls_name):
    """The fused steps cache a = L'(x - mu) for the state they wrote last.  A write into the state by the LIBRARY (here
    omc_chain_copy, which torch's version counter does not see), a fresh state tensor, and a sampler object taken into another
    run must all be noticed: the whitened sampler has to stay equal to the route that recomputes everything."""
    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.model import Model
    from openmcmc_amd.sampler import metropolis_hastings as mh

    d, C, steps = 24, 6, 9
    rng = np.random.default_rng(5)
    A = rng.standard_normal((d, 2 * d))
    Qh = np.linalg.inv(A @ A.T / (2 * d))
    Qh = (Qh + Qh.T) / 2
    mu = rng.standard_normal((d, 1))
    x0 = rng.standard_normal((C, d, 1))
    other = rng.standard_normal((C, d))
    zs, us = rng.standard_normal((steps, C, d)), rng.random((steps, C))
    outs = []
    for trust_cache in (False, True):
        eng = make_engine(C, seed=2)
        mdl = Model([Normal("x", mean="mu", precision="Q")])
        smp = getattr(mh, cls_name)("x", mdl, step=np.array([[0.5 if cls_name == "ManifoldMALA" else 0.1]]))
        smp.bind(eng)
        smp.inject = lambda s, sweep, *a: eng.to_device(zs[sweep])
        smp.inject_uniform = lambda s, sweep, *a: eng.to_device(us[sweep])
        state = {"x": ChainArray(eng.to_device(x0)), "mu": mu, "Q": Qh}
        used_cache = 0
        for i in range(steps):
            if i == 3:  # the library itself overwrites the state in place
                eng.chain_copy(eng.to_device(other), state["x"].vector())
            if i == 6:  # a fresh state entry (possibly on a recycled address, version 0 again)
                state["x"] = ChainArray(eng.to_device(x0))
            if not trust_cache:
                smp._white_tag = None  # the reference: a = L'(x - mu) recomputed from x in every step
            else:
                x = state["x"].vector()
                LQ = smp._plan[0] if smp._plan is not None else None
                if LQ is not None:
                    tag = smp._white_state_tag(state, x, LQ, eng.shared(mu).reshape(-1))
                    current = smp._white_state_is_current(eng, x, tag)
                    assert current == (i not in (0, 3, 6)), i
                    used_cache += int(current)
            state = smp.sample(state)
        eng.check_status()
        if trust_cache:
            assert used_cache == steps - 3
        outs.append((state["x"].data.cpu().numpy().copy(), smp.accept_rate.accept.cpu().numpy().copy()))
        eng.close()
    assert np.array_equal(outs[0][1], outs[1][1])
    assert relerr(outs[1][0], outs[0][0]) < 1e-10


@pytest.mark.parametrize("kind", ["mala", "rw"])
def test_whitened_steps_are_in_the_write_log(kind):
    """omc_mala_step_white / omc_rw_step_white write x in place where torch's version counter does not see it: a quadratic form
    cached from x (Engine.quad_cache_*, what a Normal-Gamma block further down the sweep would read) must not survive them."""
    d, C = 12, 5
    rng = np.random.default_rng(11)
    A = rng.standard_normal((d, 2 * d))
    Qh = np.linalg.inv(A @ A.T / (2 * d))
    Qh = (Qh + Qh.T) / 2
    eng = make_engine(C, seed=4)
    x = eng.to_device(rng.standard_normal((C, d)))
    scale = 4.0 if kind == "mala" else 1.0
    L, sl = eng.dense_cholesky(eng.to_device(Qh), scale)
    holder = object()
    eng.quad_cache_put(holder, eng.empty(1, C), [x])
    assert eng.quad_cache_get(holder, [x]) is not None
    serial = eng._write_serial
    version = x._version
    if kind == "mala":
        eng.mala_step_white(None, L, sl, 0.5, x, draw_index=1)
    else:
        eng.rw_step_white(None, L, sl, 0.1, x, draw_index=1)
    eng.check_status()
    assert x._version == version  # torch saw nothing ...
    assert eng.written_since(x, serial)  # ... the library's own log did
    assert eng.quad_cache_get(holder, [x]) is None
    eng.close()


@pytest.mark.parametrize("d,C,steps,inject", [(5, 3, 7, True), (500, 33, 70, False), (137, 70, 40, True), (1100, 5, 9, False),
                                              (500, 140, 37, False), (137, 131, 33, False)])
def test_whitened_mala_run_is_the_single_steps_in_a_row(d, C, steps, inject):
    """omc_mala_run_white (a block of steps per launch, one product per block into the store) against `steps` calls of
    omc_mala_step_white on the same draws or streams: stored states, state left behind, counters and log densities bit for bit."""
    import torch

    rng = np.random.default_rng(d + steps)
    A = rng.standard_normal((d, 2 * d))
    Qh = np.linalg.inv(A @ A.T / (2 * d))
    Qh = (Qh + Qh.T) / 2
    mu = rng.standard_normal(d)
    step = 0.5 if d <= 500 else 0.3
    x0 = mu + np.linalg.solve(np.linalg.cholesky(Qh).T, rng.standard_normal((d, C))).T  # a draw from the target
    zs = rng.standard_normal((steps, C, d)) if inject else None
    us = rng.random((steps, C)) if inject else None
    eng = make_engine(C, seed=21)
    L, sl = eng.dense_cholesky(eng.to_device(Qh), 1.0 / step**2)
    dmu = eng.to_device(mu)
    # the single steps
    x = eng.to_device(x0)
    acc, prop = (torch.zeros(C, dtype=torch.int64, device="cuda") for _ in range(2))
    lp = eng.empty(C)
    want_x, want_lp, flags = [], [], []
    for i in range(steps):
        before = acc.clone()
        eng.mala_step_white(dmu, L, sl, step, x, state_is_current=i > 0, z=None if zs is None else eng.to_device(zs[i]),
                            u=None if us is None else eng.to_device(us[i]), draw_index=100 + 3 * i, accept_count=acc,
                            proposal_count=prop, log_p_out=lp)
        want_x.append(x.cpu().numpy().copy())
        want_lp.append(lp.cpu().numpy().copy())
        flags.append((acc - before).cpu().numpy())
    # the run
    x2 = eng.to_device(x0)
    acc2, prop2 = (torch.zeros(C, dtype=torch.int64, device="cuda") for _ in range(2))
    xs, lps, lp2 = eng.empty(steps, C, d), eng.empty(steps, C), eng.empty(C)
    eng.mala_run_white(dmu, L, sl, step, x2, steps, z=None if zs is None else eng.to_device(zs), u=None if us is None else eng.to_device(us),
                       draw_index0=100, draw_stride=3, x_store=xs, logp_store=lps, accept_count=acc2, proposal_count=prop2, log_p_out=lp2)
    eng.check_status()
    assert np.array_equal(acc2.cpu().numpy(), acc.cpu().numpy()) and np.array_equal(prop2.cpu().numpy(), prop.cpu().numpy())
    assert 0 < acc.sum().item() < steps * C  # both outcomes occur
    assert np.array_equal(lps.cpu().numpy(), np.array(want_lp))
    assert np.array_equal(lp2.cpu().numpy(), want_lp[-1])
    got, want = xs.cpu().numpy(), np.array(want_x)
    # From a chain's first accepted proposal on, both routes hold x = mu + L^-T a of the same a through the same product, column
    # by column: bit-equal.  Before it the single steps still hold the caller's x0 untouched, the run holds mu + L^-T L'(x0 - mu):
    # equal to rounding.
    moved = np.cumsum(np.array(flags), axis=0) > 0  # (steps, C)
    assert moved.any()
    assert d < 100 or not moved.all()
    if 32 * C < 4096:  # (blocks of 4096 columns and more go through the 64 x 64-tile product: another summation order)
        assert np.array_equal(got[moved], want[moved])
    assert relerr(got, want) < 1e-12
    assert relerr(x2.cpu().numpy(), want_x[-1]) < 1e-12
    # without a store: only the last state comes back, and it is the same one
    x3 = eng.to_device(x0)
    eng.mala_run_white(dmu, L, sl, step, x3, steps, z=None if zs is None else eng.to_device(zs), u=None if us is None else eng.to_device(us),
                       draw_index0=100, draw_stride=3)
    assert relerr(x3.cpu().numpy(), want_x[-1]) < 1e-12
    eng.close()


def test_manifold_mala_on_eighty_regression_coefficients(golden):
    """The same structure as test_manifold_mala_on_regression_coefficients beyond one wave's worth of coefficients (p = 80 > 64: the
    per-chain Hessian combination goes through the dense route): 25 steps of the reference (tests/golden/mala_wide.npz, made by
    tests/golden/make_golden_r4.py), same accept decisions, states to 1e-9."""
    import torch
    from scipy import sparse

    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix
    from openmcmc_amd.sampler.metropolis_hastings import ManifoldMALA

    G = golden("mala_wide")
    X, w, y = G["X"], G["w"], G["y"]
    n, p = X.shape
    C = 3
    eng = make_engine(C)
    mdl = Model([Normal("y", mean=LinearCombination({"beta": "X"}), precision=ScaledMatrix("P_tau", "tau")),
                 Normal("beta", mean="mu", precision=ScaledMatrix("P_lam", "lam"))])
    dev = eng.device
    full = lambda v: ChainArray(eng.full((C, 1, 1), float(v)))  # noqa: E731
    state = {"y": y.reshape(n, 1), "X": X, "beta": ChainArray(eng.to_device(np.tile(G["beta0"], (C, 1)))),
             "P_tau": sparse.diags(w, format="csc"), "tau": full(G["tau"]), "P_lam": G["P"], "lam": full(G["lam"]),
             "mu": np.full((p, 1), float(G["mu"]))}
    smp = ManifoldMALA("beta", mdl, step=np.array(float(G["step"]))).bind(eng)
    smp.inject = lambda s, it: torch.as_tensor(np.tile(G["z"][it], (C, 1)), device=dev)
    smp.inject_uniform = lambda s, it: torch.full((C,), float(G["u"][it]), dtype=torch.float64, device=dev)
    for it in range(int(G["n_steps"])):
        before = smp.accept_rate.accept.clone()
        state = smp.sample(state)
        eng.check_status()
        got = state["beta"].numpy()[:, :, 0]
        ref = G["x"][it]
        assert np.max(np.abs(got - ref[None, :])) < 1e-9 * max(1.0, np.abs(ref).max()), it
        assert (smp.accept_rate.accept - before).cpu().numpy().tolist() == [int(G["accept"][it])] * C, it
    assert 0 < G["accept"].sum() < G["n_steps"]
    eng.close()


def test_generic_manifold_mala_beyond_one_waves_columns(golden):
    """The generic route (a Hessian that depends on the sampled vector: one Lambda_c = H_c / step^2 per chain and step) at d = 70,
    beyond the small-matrix kernels' 64 columns: 15 steps of the reference on a LogNormal-prior vector under a regression
    likelihood (tests/golden/mala_lognormal_wide.npz), decisions identical, states to 1e-8 (metropolis_hastings.py:325-348 has no
    size limit)."""
    from scipy import sparse

    from openmcmc_amd.chains import ChainArray
    from openmcmc_amd.distribution.location_scale import LogNormal, Normal
    from openmcmc_amd.model import Model
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix
    from openmcmc_amd.sampler.metropolis_hastings import ManifoldMALA

    G = golden("mala_lognormal_wide")
    C, n_obs = 3, G["A"].shape[0]
    eng = make_engine(C, seed=1)
    mdl = Model([Normal("y", mean=LinearCombination(form={"s": "A"}), precision=ScaledMatrix(matrix="P_y", scalar="tau")),
                 LogNormal("s", mean="mu_s", precision="Q_s")])
    state = {"y": G["y"].reshape(-1, 1), "A": G["A"], "s": ChainArray(eng.to_device(np.tile(G["s0"], (C, 1)))),
             "tau": ChainArray(eng.full((C, 1, 1), float(G["tau"]))), "P_y": sparse.csc_matrix(np.eye(n_obs)),
             "mu_s": G["mu_s"].reshape(-1, 1), "Q_s": G["Q_s"]}
    smp = ManifoldMALA("s", mdl, step=np.array([float(G["step"])]))
    smp.bind(eng, 0, 1)
    smp.inject = lambda s_, t: eng.to_device(np.tile(G["z"][t], (C, 1)))
    smp.inject_uniform = lambda s_, t: eng.full((C,), G["u"][t])
    flags = []
    for t in range(int(G["n_steps"])):
        before = smp.accept_rate.accept.clone()
        state = smp.sample(state)
        flags.append((smp.accept_rate.accept - before).cpu().numpy())
        assert relerr(state["s"].data[2, :, 0].cpu().numpy(), G["x"][t]) < 1e-8, t
    eng.check_status()
    flags = np.array(flags)
    for c in range(C):
        assert np.array_equal(flags[:, c], G["accept"])
    assert 0 < G["accept"].sum() < G["n_steps"]
    eng.close()
