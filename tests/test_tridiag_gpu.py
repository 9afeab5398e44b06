"""Parity of the HIP tridiagonal GMRF path (through the C ABI) against the CPU oracle and the
golden vectors of the reference.  Tolerance for Gaussian quantities: 1e-10 relative (north_star)."""

import numpy as np
import pytest
from scipy import sparse

from oracle import gmrf_ref, sweep_ref

pytestmark = pytest.mark.gpu

TOL = 1e-10


@pytest.fixture(scope="module")
def torch():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch


def make_engine(C, **kw):
    from openmcmc_amd.engine import Engine

    return Engine(C, **kw)


def relerr(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def tridiag(diag, off):
    n = diag.size
    if n == 1:
        return sparse.csc_matrix(np.array([[diag[0]]]))
    return sparse.diags((off, diag, off), offsets=[-1, 0, 1], format="csc")


def rw1(n, bump=1e-3):
    d = np.full(n, 2.0)
    d[0] = d[-1] = 1.0
    if n == 1:
        d[0] = 1.0
    d[0] += bump
    return d, -np.ones(max(n - 1, 0))


def oracle_draw(n, pd, po, lam, tau, y, mu, z, rhs_extra=None):
    """Oracle for one chain: Q = lam*P + tau*I, b = lam*P mu + tau*y (+ extra)."""
    P = tridiag(pd, po)
    Q = (lam * P + tau * sparse.identity(n, format="csc")).tocsc()
    b = lam * (P @ mu.reshape(n, 1)) + tau * y.reshape(n, 1)
    if rhs_extra is not None:
        b = b + rhs_extra.reshape(n, 1)
    x, m, L = gmrf_ref.draw_canonical(b, Q, z)
    r1, r2 = x - mu.reshape(n, 1), x - y.reshape(n, 1)
    quad = np.array([(r1.T @ P @ r1).item(), (r2.T @ r2).item()])
    logdet = 2 * np.sum(np.log(L.diagonal()))
    return np.asarray(x).ravel(), np.asarray(m).ravel(), quad, logdet


def run_case(torch, n, C, algo, seg, rng, lam_tau=None, with_extra=False):
    eng = make_engine(C)
    eng.set_option("tridiag_algo", algo)
    eng.set_option("tridiag_seg", seg)
    pd, po = rw1(n)
    pd = pd * (1 + 0.1 * rng.random(n))  # irregular diagonal, still PD after tau*I
    y = rng.standard_normal(n) + 2
    mu = 0.3 * rng.standard_normal(n)
    if lam_tau is None:
        lam = 50 + 100 * rng.random(C)
        tau = 0.5 + rng.random(C)
    else:
        lam, tau = np.full(C, lam_tau[0]) * (1 + 0.01 * rng.random(C)), np.full(C, lam_tau[1])
    z = rng.standard_normal((C, n))
    extra = rng.standard_normal((C, n)) if with_extra else None
    d_pd, d_po, d_y, d_mu = (eng.to_device(v) for v in (pd, po if n > 1 else np.zeros(1), y, mu))
    Pmu = eng.tridiag_matvec(n, d_pd, d_po if n > 1 else None, d_mu)
    terms = [
        {"diag": d_pd, "off": d_po if n > 1 else None, "rhs": Pmu, "center": d_mu, "scale": eng.to_device(lam)},
        {"rhs": d_y, "center": d_y, "scale": eng.to_device(tau)},
    ]
    x, mean = eng.empty(C, n), eng.empty(C, n)
    quad, logdet = eng.empty(2, C), eng.empty(C)
    eng.tridiag_sample_canonical(n, terms, x, z=eng.to_device(z), rhs_chain=None if extra is None else eng.to_device(extra),
                                 mean_out=mean, quad_out=quad, logdet_out=logdet)
    eng.check_status()
    x, mean, quad, logdet = (t.cpu().numpy() for t in (x, mean, quad, logdet))
    # separate quadform entry point on the drawn x
    quad2 = eng.empty(2, C)
    eng.tridiag_quadform(n, terms, eng.to_device(x), quad2)
    quad2 = quad2.cpu().numpy()
    worst = 0.0
    chains = range(C) if C <= 8 else list(range(0, C, max(1, C // 6))) + [C - 1]
    for c in chains:
        xo, mo, qo, ldo = oracle_draw(n, pd, po, lam[c], tau[c], y, mu, z[c], None if extra is None else extra[c])
        errs = [relerr(x[c], xo), relerr(mean[c], mo), relerr(quad[:, c], qo), relerr(logdet[c], ldo),
                relerr(quad2[:, c], qo)]
        worst = max(worst, *errs)
    eng.close()
    return worst


@pytest.mark.parametrize("n", [1, 2, 3, 8, 64, 257])
def test_golden_primitives(torch, golden, n):
    """Reference vectors: x, mu and the factor diagonal (through log det) for the golden (lam, tau)."""
    G = golden("tridiag_primitives")
    k = f"n{n}_"
    for algo, seg in ((1, 0), (2, 8), (2, 10), (2, 16), (2, 20), (2, 32)):
        eng = make_engine(2)
        eng.set_option("tridiag_algo", algo)
        eng.set_option("tridiag_seg", seg)
        lam, tau = float(G[k + "lam"]), float(G[k + "tau"])
        d_pd = eng.to_device(G[k + "P_diag"])
        d_po = eng.to_device(G[k + "P_off"]) if n > 1 else None
        # b is a free vector here: feed it as the per-chain rhs, no shared rhs terms
        terms = [{"diag": d_pd, "off": d_po, "scale": eng.full((2,), lam)}, {"scale": eng.full((2,), tau)}]
        b = eng.to_device(np.tile(G[k + "b"], (2, 1)))
        z = eng.to_device(np.tile(G[k + "z"], (2, 1)))
        x, mean, logdet = eng.empty(2, n), eng.empty(2, n), eng.empty(2)
        eng.tridiag_sample_canonical(n, terms, x, z=z, rhs_chain=b, mean_out=mean, logdet_out=logdet)
        eng.check_status()
        for c in range(2):
            assert relerr(x[c].cpu().numpy(), G[k + "x"]) < TOL
            assert relerr(mean[c].cpu().numpy(), G[k + "mu"]) < TOL
            assert relerr(logdet[c].item(), 2 * np.sum(np.log(G[k + "L_diag"]))) < TOL
        eng.close()


@pytest.mark.parametrize("algo,seg", [(1, 0), (2, 8), (2, 10), (2, 16), (2, 20), (2, 32)])
@pytest.mark.parametrize("n,C", [(1, 3), (2, 5), (7, 1), (15, 3), (16, 3), (17, 70), (33, 3), (100, 200),
                                 (512, 3), (1000, 9), (1025, 2), (4097, 3)])
def test_random_vs_oracle(torch, algo, seg, n, C):
    rng = np.random.default_rng(1000 * n + C)
    assert run_case(torch, n, C, algo, seg, rng) < TOL


@pytest.mark.parametrize("algo,seg,n", [(1, 0, 10000), (2, 16, 10000), (2, 10, 10000), (2, 20, 10000), (2, 32, 10000), (2, 8, 8000), (2, 32, 16384),
                                        (0, 0, 5000), (0, 0, 20000)])
def test_long_chains(torch, algo, seg, n):
    rng = np.random.default_rng(n + seg)
    assert run_case(torch, n, 3, algo, seg, rng, with_extra=True) < TOL


@pytest.mark.parametrize("lam_tau,tol", [((1e4, 1.0), 1e-10), ((1e6, 1.0), 1e-10), ((1e9, 1.0), 1e-10), ((1.0, 1e3), 1e-10), ((1e-3, 1.0), 1e-10)])
@pytest.mark.parametrize("seg", [8, 10, 16, 20, 32])
def test_weak_and_strong_coupling(torch, seg, lam_tau, tol):
    """lam/tau large = weakly coupled pivot recurrence, precisions of large magnitude.  The tolerance is the
    north-star 1e-10 throughout: tests/test_tridiag_joins_gpu.py holds oracle and kernel against a longdouble solve
    (both are good to ~1e-15 here; an earlier 1e-8 at lam/tau = 1e6 rested on a guess about the oracle)."""
    rng = np.random.default_rng(5)
    assert run_case(torch, 3000, 2, 2, seg, rng, lam_tau=lam_tau) < tol


def test_not_positive_definite(torch):
    eng = make_engine(4)
    n = 300
    pd, po = rw1(n)
    lam = np.array([10.0, 10.0, -1.0, 10.0])  # chain 2: negative definite prior block
    terms = [{"diag": eng.to_device(pd), "off": eng.to_device(po), "scale": eng.to_device(lam)},
             {"scale": eng.full((4,), 0.5)}]
    x = eng.empty(4, n)
    for algo in (1, 2):
        eng.set_option("tridiag_algo", algo)
        eng.tridiag_sample_canonical(n, terms, x, z=eng.zeros(4, n))
        with pytest.raises(np.linalg.LinAlgError, match="chain 2"):
            eng.check_status()
        eng.check_status()  # latch cleared
    eng.close()


def test_invalid_arguments(torch):
    eng = make_engine(2)
    terms = [{"scale": eng.full((2,), 1.0)}]
    with pytest.raises(ValueError):
        eng.tridiag_sample_canonical(4, terms, eng.empty(2, 3))  # x too narrow
    with pytest.raises(ValueError):
        eng.tridiag_sample_canonical(4, [{"scale": eng.full((3,), 1.0)}], eng.empty(2, 4))  # wrong chain count
    with pytest.raises(ValueError):
        eng.set_option("no_such_option", 1)
    eng.close()


@pytest.mark.parametrize("route", ["sparse", "sparsemu", "dense"])
def test_gmrf_chain_golden(torch, golden, route):
    """Replays MCMC.run_mcmc of the example-4 model (reference output in gmrf_chain.npz) through
    the ABI: NormalNormal(b) -> NormalGamma(lambda) -> NormalGamma(tau) -> store + log_post,
    with the reference's own draws injected; chain 1 of 2 carries a different start to prove
    chains do not interact."""
    G = golden("gmrf_chain")
    k = route + "_"
    n, n_burn, n_iter = int(G[k + "n"]), int(G[k + "n_burn"]), int(G[k + "n_iter"])
    C = 2
    eng = make_engine(C)
    d_pd, d_po = eng.to_device(G[k + "P_diag"]), eng.to_device(G[k + "P_off"])
    d_y, d_mu = eng.to_device(G[k + "y"]), eng.full((n,), float(G[k + "mu_val"]))
    lam, tau = eng.to_device(np.array([100.0, 37.0])), eng.to_device(np.array([1.0, 2.5]))
    Pmu = eng.tridiag_matvec(n, d_pd, d_po, d_mu)
    terms = eng.tridiag_terms([{"diag": d_pd, "off": d_po, "rhs": Pmu, "center": d_mu, "scale": lam},
                               {"rhs": d_y, "center": d_y, "scale": tau}], n)
    logdetP, logdetI = eng.tridiag_logdet(n, d_pd, d_po), eng.zeros(1)
    x, quad, lp = eng.empty(C, n), eng.empty(2, C), eng.empty(C)
    store = {key: [] for key in ("b", "lambda", "tau", "log_post")}
    for it in range(n_burn + n_iter):
        z = eng.to_device(np.tile(G[k + "z"][it], (C, 1)))
        g = G[k + "g"][it]
        eng.tridiag_sample_canonical(n, terms, x, z=z, quad_out=quad)
        eng.normal_gamma_update(10.0, 1.0, n, quad[0], lam, g=eng.full((C,), g[0]))
        eng.normal_gamma_update(1.0, 1.0, n, quad[1], tau, g=eng.full((C,), g[1]))
        if it < n_burn:
            continue
        eng.scaled_gauss_logpdf(n, tau, logdetI, quad[1], lp)
        eng.scaled_gauss_logpdf(n, lam, logdetP, quad[0], lp, accumulate=True)
        eng.gamma_logpdf(lam, 10.0, 1.0, lp, accumulate=True)
        eng.gamma_logpdf(tau, 1.0, 1.0, lp, accumulate=True)
        store["b"].append(x[0].cpu().numpy().copy())
        store["lambda"].append(lam[0].item())
        store["tau"].append(tau[0].item())
        store["log_post"].append(lp[0].item())
    eng.check_status()
    assert relerr(np.array(store["b"]).T, G[k + "store_b"]) < TOL
    assert relerr(store["lambda"], G[k + "store_lambda"].ravel()) < TOL
    assert relerr(store["tau"], G[k + "store_tau"].ravel()) < TOL
    assert relerr(store["log_post"], G[k + "store_log_post"].ravel()) < TOL
    eng.close()


@pytest.mark.parametrize("n", [5000, 10000])
def test_gmrf_big_golden(torch, golden, n):
    """BASELINE sizes: one full sweep against summaries of the reference's output."""
    G = golden("gmrf_big")
    k = f"n{n}_"
    rng = np.random.default_rng(0)
    t = np.arange(n) * 60.0 / n
    y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + rng.standard_normal(n)
    z = np.random.default_rng(int(G[k + "zseed"])).standard_normal(n)
    pd, po = rw1(n)
    C = 2
    eng = make_engine(C)
    d_pd, d_po, d_y = eng.to_device(pd), eng.to_device(po), eng.to_device(y)
    lam, tau = eng.full((C,), 100.0), eng.full((C,), 1.0)
    terms = [{"diag": d_pd, "off": d_po, "scale": lam}, {"rhs": d_y, "center": d_y, "scale": tau}]
    x, quad, lp = eng.empty(C, n), eng.empty(2, C), eng.empty(C)
    eng.tridiag_sample_canonical(n, terms, x, z=eng.to_device(np.tile(z, (C, 1))), quad_out=quad)
    g = G[k + "gdraw"]
    eng.normal_gamma_update(10.0, 1.0, n, quad[0], lam, g=eng.full((C,), g[0]))
    eng.normal_gamma_update(1.0, 1.0, n, quad[1], tau, g=eng.full((C,), g[1]))
    eng.scaled_gauss_logpdf(n, tau, eng.zeros(1), quad[1], lp)
    eng.scaled_gauss_logpdf(n, lam, eng.tridiag_logdet(n, d_pd, d_po), quad[0], lp, accumulate=True)
    eng.gamma_logpdf(lam, 10.0, 1.0, lp, accumulate=True)
    eng.gamma_logpdf(tau, 1.0, 1.0, lp, accumulate=True)
    eng.check_status()
    xh = x[1].cpu().numpy()
    assert relerr(xh[:8], G[k + "x_head"]) < TOL and relerr(xh[-8:], G[k + "x_tail"]) < TOL
    assert relerr(xh[::97], G[k + "x_stride"]) < TOL
    assert relerr(xh.sum(), G[k + "x_sum"]) < TOL and relerr((xh * xh).sum(), G[k + "x_sumsq"]) < TOL
    assert relerr(lam[1].item(), G[k + "lambda"]) < TOL and relerr(tau[1].item(), G[k + "tau"]) < TOL
    assert relerr(lp[1].item(), G[k + "log_post"]) < TOL
    eng.close()


def test_bench_size_properties(torch):
    """cfg3 at full size (1024 chains x 10 000 nodes): size-independent checks.
    (1) residual: Q x - b = L z  =>  (Qx-b)'Q^{-1}(Qx-b) = z'z, checked through the identity
        x'Qx - 2 x'b + mu'b = z'z  with mu = Q^{-1}b from the same kernel;
    (2) serial and segmented kernels agree to 1e-10 on every chain;
    (3) linearity: doubling b and z doubles x."""
    n, C = 10000, 1024
    rng = np.random.default_rng(42)
    eng = make_engine(C)
    pd, po = rw1(n)
    t = np.arange(n) * 60.0 / n
    y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + rng.standard_normal(n)
    d_pd, d_po, d_y = eng.to_device(pd), eng.to_device(po), eng.to_device(y)
    lam = eng.to_device(50 + 100 * rng.random(C))
    tau = eng.to_device(0.5 + rng.random(C))
    terms = [{"diag": d_pd, "off": d_po, "scale": lam}, {"rhs": d_y, "center": d_y, "scale": tau}]
    z = eng.fill_normal(n, draw_index=3)
    xs, x1, x2 = eng.empty(C, n), eng.empty(C, n), eng.empty(C, n)
    eng.set_option("tridiag_algo", 1)
    eng.tridiag_sample_canonical(n, terms, xs, z=z)
    eng.set_option("tridiag_algo", 2)
    q1, mean = eng.empty(2, C), eng.empty(C, n)
    eng.tridiag_sample_canonical(n, terms, x1, z=z, quad_out=q1, mean_out=mean)
    eng.check_status()
    scale = xs.abs().max().item()
    assert (xs - x1).abs().max().item() / scale < TOL
    # in-kernel generated draws must equal fill_normal's for the same draw_index
    eng.tridiag_sample_canonical(n, terms, x2, z=None, draw_index=3)
    assert (x2 - x1).abs().max().item() / scale < 1e-12
    # (1) energy identity, in torch fp64 on device (plumbing ops only)
    P_x = lambda v: d_pd * v + torch.nn.functional.pad(d_po * v[:, 1:], (0, 1)) + torch.nn.functional.pad(d_po * v[:, :-1], (1, 0))  # noqa: E731
    Qx = lam[:, None] * P_x(x1) + tau[:, None] * x1
    b = tau[:, None] * d_y[None, :]
    lhs = (x1 * Qx).sum(1) - 2 * (x1 * b).sum(1) + (mean * b).sum(1)
    zz = (z * z).sum(1)
    assert ((lhs - zz).abs() / zz).max().item() < 1e-9
    # quad outputs equal direct evaluation
    r = x1 - d_y[None, :]
    assert (((r * r).sum(1) - q1[1]).abs() / q1[1]).max().item() < TOL
    assert (((x1 * P_x(x1)).sum(1) - q1[0]).abs() / q1[0]).max().item() < TOL
    # (3) linearity
    terms2 = [{"diag": d_pd, "off": d_po, "scale": lam}, {"rhs": 2 * d_y, "center": d_y, "scale": tau}]
    eng.tridiag_sample_canonical(n, terms2, x2, z=2 * z)
    assert (x2 - 2 * x1).abs().max().item() / scale < TOL
    eng.check_status()
    eng.close()


@pytest.mark.parametrize("route", ["sparse", "sparsemu"])
@pytest.mark.parametrize("algo", [1, 2])
def test_fused_sweep_golden(torch, golden, route, algo):
    """The same replay through omc_gmrf_sweep: one launch per sweep does the draw, both
    Normal-Gamma updates, the store writes and log_post."""
    G = golden("gmrf_chain")
    k = route + "_"
    n, n_burn, n_iter = int(G[k + "n"]), int(G[k + "n_burn"]), int(G[k + "n_iter"])
    C = 3
    eng = make_engine(C)
    eng.set_option("tridiag_algo", algo)
    d_pd, d_po = eng.to_device(G[k + "P_diag"]), eng.to_device(G[k + "P_off"])
    d_y, d_mu = eng.to_device(G[k + "y"]), eng.full((n,), float(G[k + "mu_val"]))
    lam, tau = eng.to_device(np.array([100.0, 37.0, 100.0])), eng.to_device(np.array([1.0, 2.5, 1.0]))
    Pmu = eng.tridiag_matvec(n, d_pd, d_po, d_mu)
    terms = eng.tridiag_terms([{"diag": d_pd, "off": d_po, "rhs": Pmu, "center": d_mu, "scale": lam},
                               {"rhs": d_y, "center": d_y, "scale": tau}], n)
    logdetP, logdetI = eng.tridiag_logdet(n, d_pd, d_po), eng.zeros(1)
    store_b, store_lam = eng.empty(n_iter, C, n), eng.empty(n_iter, C)
    store_tau, store_lp = eng.empty(n_iter, C), eng.empty(n_iter, C)
    scratch = eng.empty(C, n)
    for it in range(n_burn + n_iter):
        j = it - n_burn
        g = G[k + "g"][it]
        blocks = [{"a0": 10.0, "b0": 1.0, "n_pos": n, "g": eng.full((C,), g[0]), "logdet": logdetP,
                   "store": store_lam[j] if j >= 0 else None},
                  {"a0": 1.0, "b0": 1.0, "n_pos": n, "g": eng.full((C,), g[1]), "logdet": logdetI,
                   "store": store_tau[j] if j >= 0 else None}]
        eng.gmrf_sweep(n, terms, blocks, store_b[j] if j >= 0 else scratch,
                       z=eng.to_device(np.tile(G[k + "z"][it], (C, 1))),
                       log_post_out=store_lp[j] if j >= 0 else None)
    eng.check_status()
    for c in (0, 2):
        assert relerr(store_b[:, c, :].cpu().numpy().T, G[k + "store_b"]) < TOL
        assert relerr(store_lam[:, c].cpu().numpy(), G[k + "store_lambda"].ravel()) < TOL
        assert relerr(store_tau[:, c].cpu().numpy(), G[k + "store_tau"].ravel()) < TOL
        assert relerr(store_lp[:, c].cpu().numpy(), G[k + "store_log_post"].ravel()) < TOL
    eng.close()


@pytest.mark.parametrize("n,C,seg", [(700, 5, 0), (10000, 64, 16), (10000, 64, 20), (10000, 64, 10), (37, 130, 0)])
def test_fused_equals_unfused(torch, n, C, seg):
    """omc_gmrf_sweep == omc_tridiag_sample_canonical + 2x omc_normal_gamma_update + the log-density
    pieces, bit for bit, with the in-kernel random streams (same draw indices)."""
    rng = np.random.default_rng(n)
    pd, po = rw1(n)
    y = rng.standard_normal(n) + 2
    out = []
    for fused in (False, True):
        eng = make_engine(C, seed=9)
        eng.set_option("tridiag_seg", seg)
        d_pd, d_po, d_y = eng.to_device(pd), eng.to_device(po), eng.to_device(y)
        lam, tau = eng.to_device(80 + np.arange(C) * 0.5), eng.full((C,), 1.25)
        terms = eng.tridiag_terms([{"diag": d_pd, "off": d_po, "scale": lam},
                                   {"rhs": d_y, "center": d_y, "scale": tau}], n)
        logdetP, logdetI = eng.tridiag_logdet(n, d_pd, d_po), eng.zeros(1)
        x, lp, quad = eng.empty(C, n), eng.empty(C), eng.empty(2, C)
        for it in range(3):
            if fused:
                blocks = [{"a0": 10.0, "b0": 1.0, "n_pos": n, "logdet": logdetP},
                          {"a0": 1.0, "b0": 1.0, "n_pos": n, "logdet": logdetI}]
                eng.gmrf_sweep(n, terms, blocks, x, draw_index=3 * it, log_post_out=lp, gamma_draw_base=3 * it + 1)
            else:
                eng.tridiag_sample_canonical(n, terms, x, draw_index=3 * it, quad_out=quad)
                eng.normal_gamma_update(10.0, 1.0, n, quad[0], lam, draw_index=3 * it + 1)
                eng.normal_gamma_update(1.0, 1.0, n, quad[1], tau, draw_index=3 * it + 2)
                eng.scaled_gauss_logpdf(n, tau, logdetI, quad[1], lp)
                eng.scaled_gauss_logpdf(n, lam, logdetP, quad[0], lp, accumulate=True)
                eng.gamma_logpdf(lam, 10.0, 1.0, lp, accumulate=True)
                eng.gamma_logpdf(tau, 1.0, 1.0, lp, accumulate=True)
        eng.check_status()
        out.append([t.cpu().numpy().copy() for t in (x, lam, tau, lp)])
        eng.close()
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert relerr(out[1][3], out[0][3]) < 1e-13  # log_post: summation order of the four terms differs


# ---------------------------------------------------------------------------------------------------
# The structure-specialised instantiation (two-term smoother: scaled identity around y + zero-mean
# tridiagonal prior, BASELINE configs[2]) against the oracle and against the generic instantiation.
def _smoother_terms(eng, n, rng, C, p_first):
    pd, po = rw1(n)
    pd = pd * (1 + 0.1 * rng.random(n))
    y = rng.standard_normal(n) + 2
    lam, tau = 50 + 100 * rng.random(C), 0.5 + rng.random(C)
    d_pd, d_po, d_y = eng.to_device(pd), eng.to_device(po), eng.to_device(y)
    tp = {"diag": d_pd, "off": d_po, "scale": eng.to_device(lam)}
    ti = {"rhs": d_y, "center": d_y, "scale": eng.to_device(tau)}
    return (pd, po, y, lam, tau), ([tp, ti] if p_first else [ti, tp])


@pytest.mark.parametrize("p_first", [True, False])
@pytest.mark.parametrize("n", [513, 4097, 4104, 5000, 8192, 8193, 9601, 10000, 10240])
def test_smoother_specialisation_vs_oracle(torch, n, p_first):
    """Injected draws: x, mean and both quadratic forms of the specialised kernel equal the oracle's; sizes
    cover both segment widths, a last wave with one node, and a chain that fills its workgroup exactly."""
    rng = np.random.default_rng(n + int(p_first))
    C = 3
    eng = make_engine(C)
    (pd, po, y, lam, tau), terms = _smoother_terms(eng, n, rng, C, p_first)
    z = rng.standard_normal((C, n))
    extra = rng.standard_normal((C, n))
    for with_extra in (False, True):
        x, mean, quad = eng.empty(C, n), eng.empty(C, n), eng.empty(2, C)
        eng.tridiag_sample_canonical(n, terms, x, z=eng.to_device(z), rhs_chain=eng.to_device(extra) if with_extra else None,
                                     mean_out=mean, quad_out=quad)
        eng.check_status()
        x, mean, quad = x.cpu().numpy(), mean.cpu().numpy(), quad.cpu().numpy()
        for c in range(C):
            xo, mo, qo, _ = oracle_draw(n, pd, po, lam[c], tau[c], y, np.zeros(n), z[c], extra[c] if with_extra else None)
            q = qo if p_first else qo[::-1]
            assert relerr(x[c], xo) < TOL and relerr(mean[c], mo) < TOL
            assert relerr(quad[:, c], q) < TOL
    eng.close()


@pytest.mark.parametrize("with_offsets", [False, True])
@pytest.mark.parametrize("n,C", [(4104, 5), (8191, 4), (10000, 6)])
def test_smoother_specialisation_equals_generic(torch, n, C, with_offsets):
    """In-kernel draws: the specialised instantiation makes its draws ahead of the forward pass (parked in
    LDS) -- the stream positions must be the ones the generic instantiation uses, so x agrees to rounding.
    With per-chain offsets the right-hand side takes the general fill; the draws must be made ahead all the same
    (an earlier version left the later pairs ungenerated on that path)."""
    rng = np.random.default_rng(7 * n)
    extra = rng.standard_normal((C, n))
    out = []
    for generic in (0, 1):
        eng = make_engine(C, seed=99)
        eng.set_option("tridiag_generic", generic)
        rs = np.random.default_rng(n)
        _, terms = _smoother_terms(eng, n, rs, C, True)
        x, quad = eng.empty(C, n), eng.empty(2, C)
        eng.tridiag_sample_canonical(n, terms, x, z=None, draw_index=5, quad_out=quad,
                                     rhs_chain=eng.to_device(extra) if with_offsets else None)
        eng.check_status()
        out.append((x.cpu().numpy(), quad.cpu().numpy()))
        eng.close()
    assert relerr(out[0][0], out[1][0]) < 1e-12
    assert relerr(out[0][1], out[1][1]) < 1e-12


def test_fused_smoother_sweeps_are_deterministic(torch):
    """The specialised kernel parks data in LDS by LDS-DMA under running arithmetic; a DMA issued before the
    wave's pending LDS reads of that region have been served showed up as rare run-to-run differences (about
    one chain in a thousand sweeps of 1024 chains).  Same state, same seed: every stored value must repeat bit
    for bit."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import GmrfSweep

    ref = None
    for _ in range(3):
        sw = GmrfSweep(10000, 1024, seed=11, chain_offset=0, device=0, n_store=4)
        sw.run_fused(1500)
        sw.eng.check_status()
        out = [t.cpu().numpy().copy() for t in (sw.store_lam, sw.store_tau, sw.store_b[0])]
        del sw
        torch.cuda.synchronize()
        if ref is None:
            ref = out
        else:
            for a, b in zip(out, ref):
                assert np.array_equal(a, b)


@pytest.mark.parametrize("misaligned", [False, True])
@pytest.mark.parametrize("with_offsets", [False, True])
@pytest.mark.parametrize("n", [4104, 10000])
def test_smoother_fused_specialised_equals_generic(torch, n, with_offsets, misaligned):
    """Three fused sweeps (in-kernel Normal and Gamma draws, log_post) through the specialised and the generic
    instantiation: per-chain offsets take the general right-hand-side fill, vectors that are only 8-byte aligned
    cannot be parked by LDS-DMA -- every such variant must leave the same chain."""
    C = 6
    rng = np.random.default_rng(n + 2 * int(with_offsets) + int(misaligned))
    pd, po = rw1(n)
    pd = pd * (1 + 0.1 * rng.random(n))
    y = rng.standard_normal(n) + 2
    extra = 0.3 * rng.standard_normal((C, n))
    out = []
    for generic in (0, 1):
        eng = make_engine(C, seed=21)
        eng.set_option("tridiag_generic", generic)

        def dev(v):
            if not misaligned:
                return eng.to_device(v)
            return eng.to_device(np.concatenate([[0.0], v]))[1:]  # data pointer = base + 8 bytes

        d_pd, d_po, d_y = dev(pd), dev(po), dev(y)
        lam, tau = eng.to_device(80 + np.arange(C) * 0.5), eng.full((C,), 1.25)
        terms = eng.tridiag_terms([{"rhs": d_y, "center": d_y, "scale": tau}, {"diag": d_pd, "off": d_po, "scale": lam}], n)
        logdetP, logdetI = eng.tridiag_logdet(n, d_pd, d_po), eng.zeros(1)
        x, lp = eng.empty(C, n), eng.empty(C)
        d_extra = eng.to_device(extra) if with_offsets else None
        for it in range(3):
            blocks = [{"a0": 1.0, "b0": 1.0, "n_pos": n, "logdet": logdetI},
                      {"a0": 10.0, "b0": 1.0, "n_pos": n, "logdet": logdetP}]
            eng.gmrf_sweep(n, terms, blocks, x, rhs_chain=d_extra, draw_index=3 * it, log_post_out=lp, gamma_draw_base=3 * it + 1)
        eng.check_status()
        out.append([t.cpu().numpy().copy() for t in (x, lam, tau, lp)])
        eng.close()
    for a, b in zip(out[0], out[1]):
        assert np.all(np.isfinite(a))
        assert relerr(a, b) < 1e-10


def test_c_entry_points_take_chains_beyond_one_workgroup():
    """n > 16 384: omc_tridiag_sample_canonical, omc_gmrf_sweep and omc_gmrf_run no longer fall to the one-lane-per-chain
    kernel (28 ms per sweep at n = 20 000 x 1024 chains) but take the segmented kernels of the band route internally.
    Against the one-lane-per-chain kernel (tridiag_algo = 1) with the same injected draws / the same Philox streams: draw,
    mean, log det, quadratic forms, the Normal-Gamma updates, the stores and log_post agree to rounding."""
    from openmcmc_amd.engine import Engine

    n, C, K = 20000, 40, 3
    rng = np.random.default_rng(8)
    t = np.arange(n) * 60.0 / 10000
    y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + rng.standard_normal(n)
    d = np.full(n, 2.0)
    d[0] = d[-1] = 1.0
    d[0] += 1e-3
    off = -np.ones(n - 1)
    lam0, tau0 = 80.0 + 40 * rng.random(C), 0.5 + rng.random(C)
    z, g = rng.standard_normal((C, n)), 5000.0 + 50 * rng.random((2, C))
    res = {}
    for algo in (0, 1):
        eng = Engine(C, seed=21)
        eng.set_option("tridiag_algo", algo)
        d_y, d_d, d_off = eng.to_device(y), eng.to_device(d), eng.to_device(off)
        lam, tau = eng.to_device(lam0), eng.to_device(tau0)
        terms = eng.tridiag_terms([{"diag": d_d, "off": d_off, "scale": lam}, {"rhs": d_y, "center": d_y, "scale": tau}], n)
        x, mean, quad, ld = eng.empty(C, n), eng.empty(C, n), eng.empty(2, C), eng.empty(C)
        eng.tridiag_sample_canonical(n, terms, x, z=eng.to_device(z), mean_out=mean, quad_out=quad, logdet_out=ld)
        eng.check_status()
        out = {"x": x.cpu().numpy(), "mean": mean.cpu().numpy(), "quad": quad.cpu().numpy(), "logdet": ld.cpu().numpy()}
        # the fused sweep with injected draws
        logdetP, logdetI = eng.full((1,), -3.7), eng.zeros(1)  # any constants: they enter log_post additively
        s_lam, s_tau, lp, xs = eng.empty(C), eng.empty(C), eng.empty(C), eng.empty(C, n)
        blocks = eng.gamma_blocks([{"a0": 10.0, "b0": 1.0, "n_pos": n, "g": eng.to_device(g[0]), "store": s_lam, "logdet": logdetP},
                                   {"a0": 1.0, "b0": 1.0, "n_pos": n, "g": eng.to_device(g[1]), "store": s_tau, "logdet": logdetI}], 2)
        eng.gmrf_sweep(n, terms, blocks, xs, z=eng.to_device(z), log_post_out=lp)
        eng.check_status()
        out.update(sweep_x=xs.cpu().numpy(), lam=lam.cpu().numpy().copy(), tau=tau.cpu().numpy().copy(), s_lam=s_lam.cpu().numpy(),
                   s_tau=s_tau.cpu().numpy(), lp=lp.cpu().numpy())
        # a short run on the Philox streams
        sb, sl, st, slp = eng.empty(K, C, n), eng.empty(K, C), eng.empty(K, C), eng.empty(K, C)
        blocks_run = [{"a0": 10.0, "b0": 1.0, "n_pos": n, "store": sl, "logdet": logdetP, "draw_index": 1},
                      {"a0": 1.0, "b0": 1.0, "n_pos": n, "store": st, "logdet": logdetI, "draw_index": 2}]
        eng.gmrf_run(n, terms, blocks_run, 1, K, 1, sb, eng.empty(C, n), draw_index0=9, draws_per_sweep=3, log_post_store=slp)
        eng.check_status()
        out.update(run_b=sb.cpu().numpy(), run_lam=sl.cpu().numpy(), run_tau=st.cpu().numpy(), run_lp=slp.cpu().numpy())
        res[algo] = out
        eng.close()
    assert np.array_equal(res[0]["lam"], res[0]["s_lam"]) and np.array_equal(res[0]["tau"], res[0]["s_tau"])
    for key in res[1]:
        a, b = res[0][key], res[1][key]
        assert np.all(np.isfinite(a)), key
        err = np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))
        assert err < (1e-8 if key.startswith("run_") else 1e-10), (key, err)


@pytest.mark.parametrize("n", [650, 1000, 3333, 10000])
@pytest.mark.parametrize("kind", ["prior_mean", "response", "both"])
def test_per_chain_centres_inside_the_launch(torch, n, kind):
    """omc_tridiag_terms.center_chain (ABI 2): a term centred at a vector that differs per chain -- the prior mean of a
    hierarchical model that is itself sampled (sampler.py:181-183) or a sampled response whose mean is the parameter
    (sampler.py:185-192).  The launch forms s_k M_k c_k itself and takes the quadratic forms around c_k; checked against
    the oracle's dense statement per chain, and against the route it replaces (product vector by omc_tridiag_matvec_chain
    fed in as rhs_chain, residual by omc_chain_lincomb + omc_tridiag_quadform)."""
    C = 5
    rng = np.random.default_rng(n)
    eng = make_engine(C, seed=5)
    assert eng.tridiag_takes_center_chain(n)
    pd, po = rw1(n)
    y = rng.standard_normal(n) + 1.0
    lam, tau = 20 + 50 * rng.random(C), 0.5 + rng.random(C)
    m = 0.3 * rng.standard_normal((C, n)) + 1.0   # per-chain prior mean
    yc = y + 0.2 * rng.standard_normal((C, n))    # per-chain response
    z = rng.standard_normal((C, n))
    d_pd, d_po, d_y = eng.to_device(pd), eng.to_device(po), eng.to_device(y)
    d_m, d_yc = eng.to_device(m), eng.to_device(yc)
    t_prior = {"diag": d_pd, "off": d_po, "scale": eng.to_device(lam)}
    t_lik = {"scale": eng.to_device(tau)}
    if kind in ("prior_mean", "both"):
        t_prior["center_chain"] = d_m
    if kind in ("response", "both"):
        t_lik["center_chain"] = d_yc
    else:
        t_lik.update(rhs=d_y, center=d_y)
    x, quad = eng.empty(C, n), eng.empty(2, C)
    if kind == "both":  # one term with a per-chain centre per call
        with pytest.raises(NotImplementedError):
            eng.tridiag_sample_canonical(n, [t_prior, t_lik], x, z=eng.to_device(z), quad_out=quad)
        eng.close()
        return
    eng.tridiag_sample_canonical(n, [t_prior, t_lik], x, z=eng.to_device(z), quad_out=quad)
    eng.check_status()
    xh, qh = x.cpu().numpy(), quad.cpu().numpy()
    for c in range(C):
        mu = m[c] if kind in ("prior_mean", "both") else np.zeros(n)
        yy = yc[c] if kind in ("response", "both") else y
        xo, _, q, _ = oracle_draw(n, pd, po, lam[c], tau[c], yy, mu, z[c])
        assert relerr(xh[c], np.asarray(xo).ravel()) < TOL
        assert relerr(qh[:, c], q) < 1e-9
    # the standalone quadratic form around the same centres
    q2 = eng.empty(2, C)
    eng.tridiag_quadform(n, [t_prior, t_lik], x, q2)
    assert relerr(q2.cpu().numpy(), qh) < 1e-12
    eng.close()


def test_per_chain_centres_are_refused_where_no_kernel_takes_them(torch):
    eng = make_engine(3, seed=1)
    n = 40  # sub-wave form
    assert not eng.tridiag_takes_center_chain(n)
    pd, po = rw1(n)
    terms = [{"diag": eng.to_device(pd), "off": eng.to_device(po), "center_chain": eng.zeros(3, n)}, {}]
    with pytest.raises(NotImplementedError):
        eng.tridiag_sample_canonical(n, terms, eng.empty(3, n), z=eng.zeros(3, n))
    eng.close()


@pytest.mark.parametrize("n", [16385, 20001, 50000])
def test_logdet_of_a_chain_longer_than_one_workgroup(n):
    """omc_tridiag_logdet beyond the segmented kernel's reach (one lane, entries fetched ahead of the recurrence, the
    logarithm taken of a running product of mantissas): against the recurrence in extended precision (gmrf.py:489-520)."""
    from openmcmc_amd.engine import Engine

    rng = np.random.default_rng(n)
    off = -(0.5 + rng.random(n - 1))
    diag = 1e-3 + 0.3 * rng.random(n)
    diag[:-1] -= off
    diag[1:] -= off
    eng = Engine(1)
    got = eng.tridiag_logdet(n, eng.to_device(diag), eng.to_device(off)).cpu().numpy()[0]
    eng.check_status()
    D = np.longdouble(diag[0])
    want = np.log(D)
    for i in range(1, n):
        D = np.longdouble(diag[i]) - np.longdouble(off[i - 1]) ** 2 / D
        want += np.log(D)
    assert abs(got - float(want)) < 1e-11 * abs(float(want))
    # a matrix that is not positive definite is reported
    diag[n // 2] = -1.0
    eng.tridiag_logdet(n, eng.to_device(diag), eng.to_device(off))
    with pytest.raises(np.linalg.LinAlgError):
        eng.check_status()
    eng.close()
