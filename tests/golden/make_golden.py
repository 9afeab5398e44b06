"""Generate the golden vectors under tests/golden/ by RUNNING the reference (openMCMC v1.0.7).

Run only in the build container, where the reference is mounted read-only:

    PYTHONPATH=/root/reference/src python3 tests/golden/make_golden.py

The reference never travels to the GPU box; only the .npz files written here do.  Every
fixture holds plain data: the inputs, the random draws the reference consumed (recorded
by wrapping scipy.stats.<dist>.rvs -- the reference draws everything through those,
gmrf.py:56, sampler.py:287, metropolis_hastings.py:173,250) and the outputs it produced.

Recorded-draw convention (verified bit-equal there):
    norm.rvs(size, scale=s)      -> standard normal z, value returned z*s (+loc)
    gamma.rvs(a, scale=s)        -> standard gamma g ~ Gamma(a,1), value returned g*s
    uniform.rvs()                -> u in [0,1)
"""

import os
import sys

import numpy as np
from scipy import sparse, stats

REF_SRC = "/root/reference/src"
if REF_SRC not in sys.path:
    sys.path.insert(0, REF_SRC)

from openmcmc import gmrf  # noqa: E402
from openmcmc.distribution.distribution import Gamma  # noqa: E402
from openmcmc.distribution.location_scale import Normal  # noqa: E402
from openmcmc.mcmc import MCMC  # noqa: E402
from openmcmc.model import Model  # noqa: E402
from openmcmc.parameter import LinearCombination, ScaledMatrix  # noqa: E402
from openmcmc.sampler.metropolis_hastings import ManifoldMALA, RandomWalk  # noqa: E402
from openmcmc.sampler.sampler import NormalGamma, NormalNormal  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


class DrawRecorder:
    """Replaces scipy.stats norm/gamma/uniform .rvs by recording versions."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.normal, self.gamma, self.uniform = [], [], []
        self._saved = None

    def _norm(self, loc=0, scale=1, size=None, **_):
        z = self.rng.standard_normal(size)
        self.normal.append(np.array(z, dtype=np.float64).reshape(-1))
        return loc + z * scale

    def _gamma(self, a, loc=0, scale=1, size=None, **_):
        a = np.asarray(a, dtype=np.float64)
        g = self.rng.standard_gamma(a, size=size)
        self.gamma.append(np.array(g, dtype=np.float64).reshape(-1))
        return loc + g * scale

    def _uniform(self, loc=0, scale=1, size=None, **_):
        u = self.rng.random(size)
        self.uniform.append(np.array(u, dtype=np.float64).reshape(-1))
        return loc + u * scale

    def __enter__(self):
        self._saved = (stats.norm.rvs, stats.gamma.rvs, stats.uniform.rvs)
        stats.norm.rvs, stats.gamma.rvs, stats.uniform.rvs = self._norm, self._gamma, self._uniform
        return self

    def __exit__(self, *exc):
        stats.norm.rvs, stats.gamma.rvs, stats.uniform.rvs = self._saved

    @staticmethod
    def cat(chunks):
        return np.concatenate(chunks) if chunks else np.zeros(0)


def rw1_precision(n, bump=1e-3):
    """RW1 precision of example 4 (examples/4_GMRF_smoother.ipynb:86-88) on unit spacing."""
    P = gmrf.precision_irregular(np.arange(float(n)))
    if sparse.issparse(P):
        P = P.tolil()
        P[0, 0] = P[0, 0] + bump
        P = P.tocsc()
    else:
        P = P.astype(float)
        P[0, 0] += bump
        P = sparse.csc_matrix(P)
    return P


def band_of(M):
    M = M.toarray() if sparse.issparse(M) else np.asarray(M)
    return np.diag(M).copy(), np.diag(M, -1).copy()


# ----------------------------------------------------------------------------- G1
def gen_tridiag_primitives():
    """gmrf.sparse_cholesky / cho_solve / solve / sample_normal_canonical /
    multivariate_normal_pdf on Q = lam*P + tau*I, sparse route (gmrf.py:167-198, 321-348, 414-520)."""
    out = {}
    rng = np.random.default_rng(11)
    sizes = [1, 2, 3, 8, 64, 257]
    out["sizes"] = np.array(sizes)
    for n in sizes:
        P = rw1_precision(n)
        lam, tau = 100.0 * (1 + 0.1 * rng.random()), 1.0 + rng.random()
        Q = (lam * P + tau * sparse.identity(n, format="csc")).tocsc()
        b = rng.standard_normal((n, 1)) * 3
        z = rng.standard_normal((n, 1))
        xval = rng.standard_normal((n, 1))
        mu_pdf = rng.standard_normal((n, 1))
        L = gmrf.sparse_cholesky(Q)
        Ld = L.toarray() if sparse.issparse(L) else np.asarray(L)
        mu = np.asarray(gmrf.cho_solve((L, True), b)).reshape(n, 1)
        saved = stats.norm.rvs
        stats.norm.rvs = lambda size=None, **k: z.reshape(size)  # inject z (gmrf.py:56)
        try:
            import openmcmc.gmrf as g

            g.norm.rvs = stats.norm.rvs
            x = np.asarray(gmrf.sample_normal_canonical(b, Q)).reshape(n, 1)
        finally:
            stats.norm.rvs = saved
            g.norm.rvs = saved
        lp = gmrf.multivariate_normal_pdf(xval, mu_pdf, Q)
        pd, po = band_of(P)
        k = f"n{n}_"
        out[k + "P_diag"], out[k + "P_off"] = pd, po
        out[k + "lam"], out[k + "tau"] = lam, tau
        out[k + "b"], out[k + "z"] = b.ravel(), z.ravel()
        out[k + "L_diag"], out[k + "L_off"] = np.diag(Ld).copy(), np.diag(Ld, -1).copy()
        out[k + "mu"], out[k + "x"] = mu.ravel(), x.ravel()
        out[k + "pdf_x"], out[k + "pdf_mu"], out[k + "logpdf"] = xval.ravel(), mu_pdf.ravel(), float(lp)
    np.savez_compressed(os.path.join(OUT, "tridiag_primitives.npz"), **out)


# ----------------------------------------------------------------------------- G2
def gen_dense_primitives():
    """Dense branch of the same functions (gmrf.py:434, 462, 481) incl. replicated log-pdf."""
    out = {}
    rng = np.random.default_rng(12)
    sizes = [1, 7, 32]
    out["sizes"] = np.array(sizes)
    import openmcmc.gmrf as g

    for p in sizes:
        A = rng.standard_normal((p, 2 * p + 3))
        Q = A @ A.T / (2 * p + 3) + 0.5 * np.eye(p)
        Q = (Q + Q.T) / 2
        b = rng.standard_normal((p, 1))
        z = rng.standard_normal((p, 1))
        nrep = 3
        xr = rng.standard_normal((p, nrep))
        mu_pdf = rng.standard_normal((p, 1))
        L = gmrf.cholesky(Q)
        mu = gmrf.cho_solve((L, True), b)
        saved = g.norm.rvs
        g.norm.rvs = lambda size=None, **k: z.reshape(size)
        try:
            x = gmrf.sample_normal_canonical(b, Q)
        finally:
            g.norm.rvs = saved
        k = f"p{p}_"
        out[k + "Q"], out[k + "b"], out[k + "z"] = Q, b.ravel(), z.ravel()
        out[k + "L"], out[k + "mu"], out[k + "x"] = L, mu.ravel(), np.asarray(x).ravel()
        out[k + "pdf_x"], out[k + "pdf_mu"] = xr, mu_pdf.ravel()
        out[k + "logpdf_sum"] = float(gmrf.multivariate_normal_pdf(xr, mu_pdf, Q))
        out[k + "logpdf_obs"] = gmrf.multivariate_normal_pdf(xr, mu_pdf, Q, by_observation=True)
    np.savez_compressed(os.path.join(OUT, "dense_primitives.npz"), **out)


# ----------------------------------------------------------------------------- G3/G4
def gmrf_model(sparse_route):
    """Example-4 model (examples/4_GMRF_smoother.ipynb:163-164).  sparse_route=True swaps
    mean="b" for LinearCombination({"b": "A"}) with A = sparse identity."""
    mean = LinearCombination(form={"b": "A"}) if sparse_route else "b"
    return Model(
        [
            Normal("y", mean=mean, precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
            Normal("b", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
            Gamma("lambda", shape="a_lam", rate="b_lam"),
            Gamma("tau", shape="a_tau", rate="b_tau"),
        ]
    )


def gmrf_data(n, seed=0):
    """Synthetic data of SURVEY.md section 8d cfg3 (shape of example 4, any n)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) * 60.0 / n
    return np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + rng.standard_normal(n)


def gmrf_state(n, y, sparse_route, mu_val=0.0):
    st = {
        "y": y.copy(),
        "b": y.copy(),
        "mu": np.full(n, mu_val),
        "lambda": 100,
        "P_lambda": rw1_precision(n),
        "a_lam": 10,
        "b_lam": 1,
        "tau": 1,
        "P_tau": sparse.csc_matrix(np.eye(n)),
        "a_tau": 1,
        "b_tau": 1,
    }
    if sparse_route:
        st["A"] = sparse.identity(n, format="csc")
    return st


def gen_gmrf_chain():
    """Full MCMC.run_mcmc (mcmc.py:87-115) of the example-4 model, both routes, recorded draws."""
    out = {}
    for route, n, n_burn, n_iter, mu_val in (("dense", 24, 3, 12, 0.0), ("sparse", 50, 5, 20, 0.0), ("sparsemu", 33, 2, 10, 0.7)):
        sparse_route = route != "dense"
        y = gmrf_data(n, seed=3)
        mdl = gmrf_model(sparse_route)
        samplers = [NormalNormal("b", mdl), NormalGamma("lambda", mdl), NormalGamma("tau", mdl)]
        with DrawRecorder(100 + n) as rec:
            import openmcmc.gmrf as g
            import openmcmc.sampler.sampler as s

            g.norm.rvs, s.gamma.rvs = stats.norm.rvs, stats.gamma.rvs
            M = MCMC(gmrf_state(n, y, sparse_route, mu_val), samplers, model=mdl, n_burn=n_burn, n_iter=n_iter)
            M.run_mcmc()
        k = route + "_"
        pd, po = band_of(rw1_precision(n))
        out[k + "n"], out[k + "n_burn"], out[k + "n_iter"], out[k + "mu_val"] = n, n_burn, n_iter, mu_val
        out[k + "y"], out[k + "P_diag"], out[k + "P_off"] = y, pd, po
        out[k + "z"] = rec.cat(rec.normal).reshape(n_burn + n_iter, n)
        out[k + "g"] = rec.cat(rec.gamma).reshape(n_burn + n_iter, 2)  # [lambda, tau] per sweep
        for key in ("b", "lambda", "tau", "log_post"):
            out[k + "store_" + key] = np.asarray(M.store[key])
        out[k + "final_b"] = np.asarray(M.state["b"]).ravel()
    np.savez_compressed(os.path.join(OUT, "gmrf_chain.npz"), **out)


def gen_gmrf_big():
    """One NormalNormal draw + both NormalGamma draws + log_post at the bench sizes, sparse route:
    summaries only (first/last entries, sums), the normal draws are re-created from the seed."""
    out = {}
    for n in (5000, 10000):
        y = gmrf_data(n, seed=0)
        mdl = gmrf_model(True)
        st = gmrf_state(n, y, True)
        st = MCMC(st, [], model=mdl, n_burn=0, n_iter=1).state
        zseed = 7000 + n
        z = np.random.default_rng(zseed).standard_normal(n)
        import openmcmc.gmrf as g
        import openmcmc.sampler.sampler as s

        saved = (g.norm.rvs, s.gamma.rvs)
        gdraw = np.array([1234.5, 8765.25]) if n == 10000 else np.array([600.0, 2400.0])
        it = iter(gdraw)
        g.norm.rvs = lambda size=None, **k: z.reshape(size)
        s.gamma.rvs = lambda a, scale=1, **k: np.asarray(next(it) * scale)
        try:
            st = NormalNormal("b", mdl).sample(st)
            x = np.asarray(st["b"]).ravel().copy()
            st = NormalGamma("lambda", mdl).sample(st)
            st = NormalGamma("tau", mdl).sample(st)
            lp = mdl.log_p(st)
        finally:
            g.norm.rvs, s.gamma.rvs = saved
        k = f"n{n}_"
        out[k + "zseed"], out[k + "gdraw"] = zseed, gdraw
        out[k + "x_head"], out[k + "x_tail"] = x[:8], x[-8:]
        out[k + "x_sum"], out[k + "x_sumsq"] = x.sum(), (x * x).sum()
        out[k + "x_stride"] = x[::97].copy()
        out[k + "lambda"], out[k + "tau"] = float(st["lambda"].item()), float(st["tau"].item())
        out[k + "log_post"] = float(lp)
    np.savez_compressed(os.path.join(OUT, "gmrf_big.npz"), **out)


# ----------------------------------------------------------------------------- G5
def gen_linreg_chain():
    """Example 3 verbatim (examples/3_linear_regression.ipynb:65-71, 158-200), 200 sweeps, and a
    wider p=7 variant; recorded draws.  BASELINE.json configs[0]/[1] shape."""
    out = {}
    for tag, N, p, n_burn, n_iter in (("ex3", 100, 2, 100, 100), ("p7", 40, 7, 5, 25)):
        rng = np.random.default_rng(5 + p)
        if p == 2:
            xx = np.sort(rng.random(N))
            X = np.stack([np.ones(N), xx], 1)
            beta_true, tau_true = np.array([2, 0.5]), 100.0
        else:
            X = rng.standard_normal((N, p))
            beta_true, tau_true = rng.standard_normal(p), 100.0
        y = X @ beta_true + rng.standard_normal(N) / np.sqrt(tau_true)
        mdl = Model(
            [
                Normal("y", mean=LinearCombination(form={"beta": "X"}), precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
                Normal("beta", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
                Gamma("tau", shape="a_tau", rate="b_tau"),
                Gamma("lambda", shape="a_lambda", rate="b_lambda"),
            ],
            response={"y": "mean"},
        )
        samplers = [NormalNormal("beta", mdl), NormalGamma("tau", mdl), NormalGamma("lambda", mdl)]
        st = {
            "y": y, "X": X, "beta": [0.0] * p, "P_tau": sparse.csc_matrix(np.eye(N)), "tau": 1,
            "P_lambda": sparse.csc_matrix(np.eye(p)), "mu": [0.0] * p, "lambda": 0.01,
            "a_tau": 1e-3, "b_tau": 1e-3, "a_lambda": 1e-3, "b_lambda": 1e-3,
        }  # fmt: skip
        with DrawRecorder(77 + p) as rec:
            import openmcmc.gmrf as g
            import openmcmc.sampler.sampler as s

            g.norm.rvs, s.gamma.rvs = stats.norm.rvs, stats.gamma.rvs
            M = MCMC(st, samplers, model=mdl, n_burn=n_burn, n_iter=n_iter)
            M.run_mcmc()
        k = tag + "_"
        out[k + "N"], out[k + "p"], out[k + "n_burn"], out[k + "n_iter"] = N, p, n_burn, n_iter
        out[k + "X"], out[k + "y"] = X, y
        out[k + "z"] = rec.cat(rec.normal).reshape(n_burn + n_iter, p)
        out[k + "g"] = rec.cat(rec.gamma).reshape(n_burn + n_iter, 2)  # [tau, lambda] per sweep
        for key in ("beta", "tau", "lambda", "log_post", "y"):
            out[k + "store_" + key] = np.asarray(M.store[key])
    np.savez_compressed(os.path.join(OUT, "linreg_chain.npz"), **out)


# ----------------------------------------------------------------------------- G7
def gen_mala():
    """ManifoldMALA / RandomWalk on a dense correlated Gaussian (metropolis_hastings.py:102-173,
    212-269, 301-373): per-step internals for one proposal and a short chain trace."""
    out = {}
    import openmcmc.gmrf as g
    import openmcmc.sampler.metropolis_hastings as mh

    dims = [1, 5, 32]
    out["dims"] = np.array(dims)
    for d in dims:
        rng = np.random.default_rng(40 + d)
        A = rng.standard_normal((d, 2 * d + 2))
        Sig = A @ A.T / (2 * d + 2)
        Q = np.linalg.inv(Sig)
        Q = (Q + Q.T) / 2
        mdl = Model([Normal("x", mean="mu", precision="Q")])
        st0 = {"x": rng.standard_normal((d, 1)), "mu": np.zeros((d, 1)), "Q": Q}
        k = f"d{d}_"
        out[k + "Q"], out[k + "x0"] = Q, st0["x"].ravel()
        for name, cls, step in (("mala", ManifoldMALA, 0.5), ("rw", RandomWalk, 0.05)):
            smp = cls("x", mdl, step=np.array([[step]]))
            n_steps = 40
            with DrawRecorder(900 + d) as rec:
                g.norm.rvs, mh.norm.rvs, mh.uniform.rvs = stats.norm.rvs, stats.norm.rvs, stats.uniform.rvs
                state = {kk: np.array(v, copy=True) for kk, v in st0.items()}
                xs, acc = [], []
                for _ in range(n_steps):
                    before = smp.accept_rate.count["accept"]
                    state = smp.sample(state)
                    xs.append(state["x"].ravel().copy())
                    acc.append(smp.accept_rate.count["accept"] - before)
            out[k + name + "_step"] = step
            out[k + name + "_z"] = rec.cat(rec.normal).reshape(n_steps, d)
            out[k + name + "_u"] = rec.cat(rec.uniform).reshape(n_steps)
            out[k + name + "_x"] = np.array(xs)
            out[k + name + "_accept"] = np.array(acc, dtype=np.int64)
        # internals of one mMALA proposal from x0 with a fixed z
        smp = ManifoldMALA("x", mdl, step=np.array([[0.5]]))
        z = np.random.default_rng(77).standard_normal((d, 1))
        saved = g.norm.rvs
        g.norm.rvs = lambda size=None, **kw: z.reshape(size)
        try:
            state = {kk: np.array(v, copy=True) for kk, v in st0.items()}
            grad, hess = mdl.grad_log_p(state, "x")
            mu_cr, chol_cr = smp._proposal_params(state)
            prop, lq_f, lq_r = smp.proposal(state)
        finally:
            g.norm.rvs = saved
        lp_c, lp_p = mdl.log_p(state), mdl.log_p(prop)
        out[k + "one_z"], out[k + "one_grad"], out[k + "one_hess"] = z.ravel(), grad.ravel(), hess
        out[k + "one_mu"], out[k + "one_chol"] = mu_cr.ravel(), chol_cr
        out[k + "one_prop"] = prop["x"].ravel()
        out[k + "one_lq_fwd"], out[k + "one_lq_rev"] = float(np.squeeze(lq_f)), float(np.squeeze(lq_r))
        out[k + "one_lp_cur"], out[k + "one_lp_prop"] = float(lp_c), float(lp_p)
    np.savez_compressed(os.path.join(OUT, "mala.npz"), **out)


# ----------------------------------------------------------------------------- G9
def gen_precision_builders():
    """gmrf.precision_irregular / precision_temporal (gmrf.py:351-411)."""
    import pandas as pd

    out = {}
    rng = np.random.default_rng(21)
    for n in (1, 2, 5, 40):
        s = np.cumsum(rng.exponential(size=n))
        P = gmrf.precision_irregular(s)
        P = P.toarray() if sparse.issparse(P) else np.asarray(P, dtype=float)
        out[f"irr{n}_s"], out[f"irr{n}_P"] = s, P
    t = pd.date_range(start="2022-04-01T01:00:00", end="2022-04-01T01:01:00", periods=30)
    out["temporal_seconds"] = np.asarray((t - t[0]).total_seconds())
    out["temporal_P"] = gmrf.precision_temporal(time=t).toarray()
    out["temporal_P_unit30"] = gmrf.precision_temporal(time=t, unit_length=30.0).toarray()
    np.savez_compressed(os.path.join(OUT, "precision_builders.npz"), **out)


# ----------------------------------------------------------------------------- timings
def gen_reference_timing():
    """Single-chain rates on cfg3 (sparse route) of the REFERENCE and of the oracle's restatement, same process, same cores,
    interleaved (the ratio that carries the GPU box's oracle timing over to the reference)."""
    import json
    import time

    from scipy import sparse

    sys.path.insert(0, os.path.abspath(os.path.join(OUT, "..", "..")))
    from oracle import sweep_ref

    n, k = 10000, 30
    y = gmrf_data(n, seed=0)
    mdl = gmrf_model(True)

    def run_reference():
        samplers = [NormalNormal("b", mdl), NormalGamma("lambda", mdl), NormalGamma("tau", mdl)]
        M = MCMC(gmrf_state(n, y, True), samplers, model=mdl, n_burn=0, n_iter=k)
        t0 = time.perf_counter()
        M.run_mcmc()
        return time.perf_counter() - t0

    d = np.full(n, 2.0)
    d[0] = d[-1] = 1.0
    d[0] += 1e-3
    off = -np.ones(n - 1)
    P = sparse.diags((off, d, off), offsets=[-1, 0, 1], format="csc")
    rng = np.random.default_rng(1)
    z, g = rng.standard_normal((k, n)), rng.standard_gamma(5000.0, size=(k, 2))

    def run_oracle():
        t0 = time.perf_counter()
        sweep_ref.gmrf_smoother_chain(y.ravel(), P, 0, k, z, g)
        return time.perf_counter() - t0

    run_reference(), run_oracle()  # warm both (imports, first-touch)
    ref, orc = [], []
    for _ in range(3):
        ref.append(run_reference())
        orc.append(run_oracle())
    t_ref, t_orc = min(ref), min(orc)
    rec = {
        "config": f"cfg3 sparse route, n={n}, 1 chain, {k} sweeps incl. store+log_post; best of 3 interleaved runs each",
        "ms_per_chain_update": 1e3 * t_ref / k,
        "chain_updates_per_s": k / t_ref,
        "oracle_ms_per_chain_update": 1e3 * t_orc / k,
        "oracle_chain_updates_per_s": k / t_orc,
        "oracle_over_reference_speed": t_ref / t_orc,
        "note": "oracle = oracle/sweep_ref.gmrf_smoother_chain (bench.py's cpu_baseline leg); reference = openmcmc MCMC.run_mcmc; "
                "reference rate on another host ~ that host's oracle rate / oracle_over_reference_speed",
        "all_runs_s": {"reference": ref, "oracle": orc},
        "cpu_count": os.cpu_count(),
        "numpy": np.__version__,
    }
    with open(os.path.join(OUT, "reference_timing.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(rec)


# ----------------------------------------------------------------------------- G8
def gen_rj_moves():
    """ReversibleJump.get_move_type / get_move_probabilities / deletion index
    (reversible_jump.py:173, 310-373): 10 000-step traces of the move bookkeeping, walking n through
    every edge case (every proposal accepted so that n visits 1, 2, n_max-1, n_max)."""
    from openmcmc.sampler.reversible_jump import ReversibleJump
    import openmcmc.sampler.reversible_jump as rj

    out = {}
    cases = [(2, 0.5, 1), (2, 0.3, 2), (3, 0.5, 2), (3, 0.3, 3), (20, 0.5, 1), (20, 0.3, 20), (20, 0.7, 19), (5, 0.5, 2)]
    out["cases"] = np.array(cases, dtype=float)
    for ci, (n_max, q, n0) in enumerate(cases):
        smp = ReversibleJump(param="n", model=Model([]), n_max=n_max, birth_probability=q)
        rng = np.random.default_rng(500 + ci)
        used_u, used_idx = [], []

        def _uniform(loc=0, scale=1, size=None, **_):
            u = rng.random()
            used_u.append(u)
            return u

        def _randint(low, high, size=None, **_):
            v = int(rng.integers(low, int(np.asarray(high).item())))
            used_idx.append(v)
            return v

        saved = (rj.uniform.rvs, rj.randint.rvs)
        rj.uniform.rvs, rj.randint.rvs = _uniform, _randint
        try:
            n = n0
            rows = []
            for _ in range(10000):
                state = {"n": np.array([[float(n)]])}
                nu = len(used_u)
                birth = bool(smp.get_move_type(state))
                pb, pd = smp.get_move_probabilities(state, birth)
                u = used_u[-1] if len(used_u) > nu else -1.0
                idx = -1
                if not birth:
                    idx = rj.randint.rvs(low=0, high=state["n"])  # reversible_jump.py:173
                rows.append((n, u, int(birth), pb, pd, idx))
                n = n + 1 if birth else n - 1
        finally:
            rj.uniform.rvs, rj.randint.rvs = saved
        out[f"case{ci}"] = np.array(rows, dtype=float)
    np.savez_compressed(os.path.join(OUT, "rj_moves.npz"), **out)


if __name__ == "__main__":
    gen_tridiag_primitives()
    gen_dense_primitives()
    gen_gmrf_chain()
    gen_gmrf_big()
    gen_linreg_chain()
    gen_mala()
    gen_precision_builders()
    gen_rj_moves()
    gen_reference_timing()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
