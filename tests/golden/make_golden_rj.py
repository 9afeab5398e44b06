"""Golden vectors for the reversible-jump rows of SURVEY.md section 8 (a12-a14, a16, cfg5), made by
RUNNING the reference (openMCMC v1.0.7) in the build container:

    PYTHONPATH=/root/reference/src python3 tests/golden/make_golden_rj.py

Writes truncnorm.npz, rj_gmrf_chain.npz and example2.npz (pass names to regenerate a subset).  Fixtures hold data only: inputs, the draws the
reference consumed, and what it produced.  Recorded-draw convention for the new kinds (SURVEY.md
section 8c, re-checked at the top of gen_truncnorm):

    truncnorm.rvs(a, b, loc, scale, size) -> u ~ U(0,1); value returned truncnorm.ppf(u, a, b)*scale + loc
    randint.rvs(low, high)                -> the integer itself
"""

import os
import sys

import numpy as np
from scipy import sparse, stats

REF_SRC = "/root/reference/src"
if REF_SRC not in sys.path:
    sys.path.insert(0, REF_SRC)

from openmcmc import gmrf, parameter  # noqa: E402
from openmcmc.distribution.distribution import Gamma, Poisson, Uniform  # noqa: E402
from openmcmc.distribution.location_scale import Normal  # noqa: E402
from openmcmc.mcmc import MCMC  # noqa: E402
from openmcmc.model import Model  # noqa: E402
from openmcmc.sampler.metropolis_hastings import MetropolisHastings, RandomWalkLoop  # noqa: E402
from openmcmc.sampler.reversible_jump import ReversibleJump  # noqa: E402
from openmcmc.sampler.sampler import NormalGamma, NormalNormal  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- truncated normal
def gen_truncnorm():
    """gmrf.truncated_normal_rv / truncated_normal_log_pdf (gmrf.py:269-318) on a grid that covers
    central, one-sided and deep-tail windows (the windows cfg5 meets: limits +-10, scales 0.2 and 1)."""
    # the draw convention: truncnorm.rvs == ppf(uniform) * scale + loc on the legacy global RandomState
    np.random.seed(5)
    got = gmrf.truncated_normal_rv(np.array([0.3, 9.9]), np.array([0.2, 0.2]), np.array([-10.0]), np.array([10.0]), size=2)
    u = np.random.RandomState(5).uniform(size=2)
    a, b = (-10 - np.array([0.3, 9.9])) / 0.2, (10 - np.array([0.3, 9.9])) / 0.2
    assert np.array_equal(got, stats.truncnorm.ppf(u, a, b) * 0.2 + np.array([0.3, 9.9]))

    rng = np.random.default_rng(31)
    rows = []
    for mean in (-9.99, -9.5, -3.0, 0.0, 0.4, 7.7, 9.9, 9.999, 12.0, -14.0, 25.0):
        for scale in (0.05, 0.2, 1.0, 3.0):
            for lower, upper in ((-10.0, 10.0), (0.5, 2.0), (-np.inf, 1.0), (-2.0, np.inf)):
                for u in (1e-12, 1e-3, 0.25, 0.5, 0.9, 1 - 1e-9, float(rng.random())):
                    rows.append((mean, scale, lower, upper, u))
    arr = np.array(rows)
    mean, scale, lower, upper, u = arr.T
    a, b = (lower - mean) / scale, (upper - mean) / scale
    x = stats.truncnorm.ppf(u, a, b) * scale + mean
    lp_fwd = gmrf.truncated_normal_log_pdf(x, mean, scale, lower, upper)
    lp_rev = gmrf.truncated_normal_log_pdf(mean, x, scale, lower, upper)  # reverse move (metropolis_hastings.py:257)
    np.savez_compressed(os.path.join(OUT, "truncnorm.npz"), mean=mean, scale=scale, lower=lower, upper=upper, u=u,
                        x=x, logpdf_fwd=lp_fwd, logpdf_rev=lp_rev)


# ----------------------------------------------------------------------------- cfg5-shaped model
def make_basis(X, knots):
    """Gaussian-kernel basis of the reference's own RJ test (tests/test_reversible_jump.py:24-40), unit scales."""
    B = np.full((X.shape[0], knots.shape[1]), np.nan)
    for k in range(knots.shape[1]):
        B[:, [k]] = stats.norm.pdf(X, loc=knots[:, k], scale=1.0)
    return B


def move_function(state, col):
    state["B"] = make_basis(state["X"], state["theta"])
    return state, 0.0, 0.0


def birth_function(cur, prop):
    prop["B"] = make_basis(prop["X"], prop["theta"])
    prop["alloc_beta"] = np.concatenate((prop["alloc_beta"], np.array([0], ndmin=2)), axis=0)
    return prop, 0.0, 0.0


def death_function(cur, prop, idx):
    prop["B"] = np.delete(prop["B"], obj=idx, axis=1)
    prop["alloc_beta"] = np.delete(prop["alloc_beta"], obj=idx, axis=0)
    return prop, 0.0, 0.0


def rj_gmrf_problem(n, n_max, seed=0):
    """Data + model + sampler list of SURVEY.md section 8d cfg5 (any n, n_max)."""
    rng = np.random.default_rng(seed)
    X = np.linspace(-10, 10, n).reshape(n, 1)
    theta_true = np.array([[-6.0, -1.0, 4.5]])
    beta_true = np.array([[3.0], [-2.0], [4.0]])
    b_true = 0.05 * np.cumsum(rng.standard_normal(n))
    y = make_basis(X, theta_true) @ beta_true + b_true.reshape(n, 1) + 0.1 * rng.standard_normal((n, 1))
    P = sparse.csc_matrix(gmrf.precision_irregular(np.arange(float(n)))).tolil()
    P[0, 0] += 1e-3
    mdl = Model(
        [
            Normal("y", mean=parameter.LinearCombination({"beta": "B", "b": "A"}), precision=parameter.ScaledMatrix("P_tau", "tau")),
            Normal("b", mean="mu_b", precision=parameter.ScaledMatrix("P_lambda", "lambda")),
            Normal("beta", mean=parameter.MixtureParameterVector("mu_beta", "alloc_beta"),
                   precision=parameter.MixtureParameterMatrix("tau_beta", "alloc_beta")),
            Poisson("n_basis", rate="rho"),
            Uniform("theta", domain_response_lower=np.array([[-10.0]]), domain_response_upper=np.array([[10.0]])),
            Gamma("lambda", shape="a_lam", rate="b_lam"),
            Gamma("tau", shape="a_tau", rate="b_tau"),
        ]
    )
    mdl.response = {"y": "mean"}
    shared = {
        "y": y, "X": X, "A": sparse.eye(n, format="csc"), "P_tau": sparse.eye(n, format="csc"), "P_lambda": P.tocsc(),
        "mu_b": np.zeros((n, 1)), "mu_beta": np.zeros((1, 1)), "tau_beta": 0.25 * np.ones((1, 1)), "rho": 5.0,
        "a_lam": 10.0, "b_lam": 1.0, "a_tau": 1.0, "b_tau": 1.0,
    }  # fmt: skip

    def samplers():
        return [
            NormalNormal("b", mdl),
            NormalNormal("beta", mdl, max_variable_size=n_max),
            NormalGamma("lambda", mdl),
            NormalGamma("tau", mdl),
            RandomWalkLoop("theta", mdl, step=np.array(0.2), max_variable_size=n_max, domain_limits=np.array([[-10.0, 10.0]]),
                           state_update_function=move_function),
            ReversibleJump("n_basis", mdl, associated_params=["theta"], n_max=n_max, state_birth_function=birth_function,
                           state_death_function=death_function,
                           matching_params={"variable": "beta", "matrix": "B", "scale": 1.0, "limits": [-10.0, 10.0]}),
        ]  # fmt: skip

    return mdl, shared, samplers


def chain_init(shared, k0, rng):
    theta = rng.uniform(-10, 10, size=(1, k0))
    st = dict(shared)
    st.update({"theta": theta, "B": make_basis(shared["X"], theta), "beta": rng.standard_normal((k0, 1)),
               "b": np.zeros_like(shared["y"]), "n_basis": k0, "alloc_beta": np.zeros((k0, 1), dtype=int),
               "lambda": 100.0, "tau": 10.0})  # fmt: skip
    return st


class Tape:
    """Per-sweep record of every draw and of the MH internals, keyed by the consuming call site."""

    def __init__(self, seed, n, n_max):
        self.rng = np.random.default_rng(seed)
        self.n, self.n_max = n, n_max
        self.rows, self.cur, self.where = [], None, None

    def new_sweep(self):
        k = self.n_max
        self.cur = {
            "z_b": np.full(self.n, np.nan), "z_beta": np.full(k, np.nan), "g": np.full(2, np.nan),
            "rw_u": np.full(k, np.nan), "rw_acc_u": np.full(k, np.nan), "rw_z": np.full(k, np.nan),
            "rw_lq_fwd": np.full(k, np.nan), "rw_lq_rev": np.full(k, np.nan), "rw_log_accept": np.full(k, np.nan),
            "rj_move_u": np.nan, "rj_theta_u": np.nan, "rj_beta_u": np.nan, "rj_idx": -1.0, "rj_acc_u": np.nan,
            "rj_birth": np.nan, "rj_lq_fwd": np.nan, "rj_lq_rev": np.nan, "rj_log_accept": np.nan,
            "rj_prop_beta": np.full(k, np.nan), "rj_prop_theta": np.full(k, np.nan), "rj_n_before": np.nan,
        }  # fmt: skip
        self.rows.append(self.cur)
        self._gi = 0

    # --- replacement draw functions
    def norm(self, loc=0, scale=1, size=None, **_):
        z = self.rng.standard_normal(size)
        flat = np.asarray(z).reshape(-1)
        if self.where == "b":
            self.cur["z_b"][:] = flat
        elif self.where == "beta":
            self.cur["z_beta"][: flat.size] = flat
        else:
            raise RuntimeError(f"unexpected normal draw in {self.where}")
        return loc + z * scale

    def gamma(self, a, loc=0, scale=1, size=None, **_):
        g = self.rng.standard_gamma(np.asarray(a, dtype=np.float64), size=size)
        self.cur["g"][0 if self.where == "lambda" else 1] = float(np.asarray(g).reshape(-1)[0])
        return loc + g * scale

    def uniform(self, loc=0, scale=1, size=None, **_):
        u = self.rng.random(size)
        if self.where == "theta":  # accept/reject of knot self.knot (metropolis_hastings.py:173)
            self.cur["rw_acc_u"][self.knot] = float(u)
        elif self.where == "n_basis":
            if size is not None:  # Uniform.rvs of the new knot (distribution.py:456)
                self.cur["rj_theta_u"] = float(np.asarray(u).reshape(-1)[0])
            elif self.stage == "move":  # get_move_type (reversible_jump.py:333)
                self.cur["rj_move_u"] = float(u)
            else:
                self.cur["rj_acc_u"] = float(u)
        else:
            raise RuntimeError(f"unexpected uniform draw in {self.where}")
        return loc + u * scale

    def truncnorm(self, a, b, loc=0, scale=1, size=None, **_):
        u = self.rng.random(size)
        if self.where == "theta":
            self.cur["rw_u"][self.knot] = float(np.asarray(u).reshape(-1)[0])
        else:
            self.cur["rj_beta_u"] = float(np.asarray(u).reshape(-1)[0])
        return stats.truncnorm.ppf(u, a, b) * scale + loc

    def randint(self, low, high, size=None, **_):
        v = int(self.rng.integers(low, int(np.asarray(high).item())))
        self.cur["rj_idx"] = float(v)
        return v


def run_reference_chain(mdl, state, samplers, tape, n_iter):
    """MCMC.run_mcmc (mcmc.py:87-115) with every sampler call wrapped so the tape knows who is drawing."""
    names = ["b", "beta", "lambda", "tau", "theta", "n_basis"]

    def wrap_sample(smp, name):
        inner = smp.sample

        def sample(st):
            if name == "b":
                tape.new_sweep()
            tape.where, tape.stage = name, "move"
            return inner(st)

        smp.sample = sample

    for smp, name in zip(samplers, names):
        wrap_sample(smp, name)

    rw, rj = samplers[4], samplers[5]
    rw_prop = rw.proposal

    def rw_proposal(st, param_index=None):
        tape.knot = param_index
        prop, f, r = rw_prop(st, param_index)
        tape.cur["rw_z"][param_index] = prop["theta"][0, param_index]
        tape.cur["rw_lq_fwd"][param_index], tape.cur["rw_lq_rev"][param_index] = float(f), float(r)
        return prop, f, r

    rw.proposal = rw_proposal
    rj_prop = rj.proposal

    def rj_proposal(st, param_index=None):
        tape.cur["rj_n_before"] = float(np.asarray(st["n_basis"]).item())
        prop, f, r = rj_prop(st)
        tape.stage = "accept"
        k = prop["beta"].shape[0]
        tape.cur["rj_birth"] = float(k > st["beta"].shape[0])
        tape.cur["rj_prop_beta"][:k] = prop["beta"].ravel()
        tape.cur["rj_prop_theta"][:k] = prop["theta"].ravel()
        tape.cur["rj_lq_fwd"], tape.cur["rj_lq_rev"] = float(np.squeeze(f)), float(np.squeeze(r))
        return prop, f, r

    rj.proposal = rj_proposal

    saved_accept = MetropolisHastings.accept_proposal

    def accept_proposal(log_accept):
        la = float(np.squeeze(log_accept))
        if tape.where == "theta":
            tape.cur["rw_log_accept"][tape.knot] = la
        else:
            tape.cur["rj_log_accept"] = la
        return saved_accept(log_accept)

    MetropolisHastings.accept_proposal = staticmethod(accept_proposal)
    saved = (stats.norm.rvs, stats.gamma.rvs, stats.uniform.rvs, stats.truncnorm.rvs, stats.randint.rvs)
    stats.norm.rvs, stats.gamma.rvs, stats.uniform.rvs = tape.norm, tape.gamma, tape.uniform
    stats.truncnorm.rvs, stats.randint.rvs = tape.truncnorm, tape.randint
    try:
        M = MCMC(state, samplers, model=mdl, n_burn=0, n_iter=n_iter)
        M.run_mcmc()
    finally:
        stats.norm.rvs, stats.gamma.rvs, stats.uniform.rvs, stats.truncnorm.rvs, stats.randint.rvs = saved
        MetropolisHastings.accept_proposal = staticmethod(saved_accept)
    return M


def gen_rj_gmrf_chain():
    """Full MCMC.run_mcmc of the cfg5-shaped model for several independent chains that share the data
    (what the chain-batched build runs in one go): the draw tape, the MH internals per sweep, the store."""
    n, n_max, n_iter = 48, 6, 150
    mdl, shared, make_samplers = rj_gmrf_problem(n, n_max, seed=2)
    out = {"n": n, "n_max": n_max, "n_iter": n_iter, "y": shared["y"].ravel(), "X": shared["X"].ravel()}
    P = shared["P_lambda"].toarray()
    out["P_diag"], out["P_off"] = np.diag(P).copy(), np.diag(P, -1).copy()
    inits = (1, 3, 5, 6, 2)
    per_chain = []
    for c, k0 in enumerate(inits):
        rng = np.random.default_rng(900 + c)
        st = chain_init(shared, k0, rng)
        init = {"theta": np.full(n_max, np.nan), "beta": np.full(n_max, np.nan)}
        init["theta"][:k0], init["beta"][:k0] = st["theta"].ravel(), st["beta"].ravel()
        tape = Tape(4000 + c, n, n_max)
        samplers = make_samplers()
        M = run_reference_chain(mdl, st, samplers, tape, n_iter)
        rec = {"init_theta": init["theta"], "init_beta": init["beta"], "init_k": float(k0)}
        for key in tape.rows[0]:
            rec["tape_" + key] = np.array([row[key] for row in tape.rows])
        for key in ("b", "beta", "lambda", "tau", "theta", "n_basis", "log_post", "y"):
            rec["store_" + key] = np.asarray(M.store[key])
        rec["accept_rw"] = np.array([samplers[4].accept_rate.count["accept"], samplers[4].accept_rate.count["proposal"]], dtype=float)
        rec["accept_rj"] = np.array([samplers[5].accept_rate.count["accept"], samplers[5].accept_rate.count["proposal"]], dtype=float)
        per_chain.append(rec)
        print("chain", c, "k0", k0, "n_basis visits", np.unique(M.store["n_basis"]), samplers[4].accept_rate.get_acceptance_rate(),
              samplers[5].accept_rate.get_acceptance_rate())
    for key in per_chain[0]:
        out[key] = np.stack([rec[key] for rec in per_chain])
    np.savez_compressed(os.path.join(OUT, "rj_gmrf_chain.npz"), **out)


# ----------------------------------------------------------------------------- example 2
def gen_example2():
    """examples/2_samplers.ipynb verbatim: five replicated observations y (1, 5) of a scalar h, samplers
    RandomWalk('h', step=5.0) and NormalNormal('h'), 300 iterations each, recorded draws."""
    from openmcmc.sampler.metropolis_hastings import RandomWalk

    def fresh():
        mdl = Model([Normal("y", mean="h", precision="tau"), Normal("h", mean="mu", precision="lambda")])
        st = {"y": np.array([150, 155, 190, 160, 173], ndmin=2), "h": np.array(200, ndmin=2),
              "tau": np.array(1 / 200, ndmin=2), "mu": np.array(160, ndmin=2), "lambda": np.array(1 / 100, ndmin=2)}
        return mdl, st

    out = {"y": np.array([150.0, 155, 190, 160, 173]), "h0": 200.0, "tau": 1 / 200, "mu": 160.0, "lambda": 1 / 100,
           "n_iter": 300, "step": 5.0}
    rng = np.random.default_rng(64)
    zs, us = [], []

    def _norm(loc=0, scale=1, size=None, **_):
        z = rng.standard_normal(size)
        zs.append(np.asarray(z, dtype=float).reshape(-1))
        return loc + z * scale

    def _uniform(loc=0, scale=1, size=None, **_):
        u = rng.random(size)
        us.append(float(u))
        return loc + u * scale

    saved = (stats.norm.rvs, stats.uniform.rvs)
    stats.norm.rvs, stats.uniform.rvs = _norm, _uniform
    try:
        mdl, st = fresh()
        smp = RandomWalk("h", model=mdl, step=5.0)
        M = MCMC(st, [smp], model=mdl, n_burn=0, n_iter=300)
        M.run_mcmc()
        out["rw_z"], out["rw_u"] = np.concatenate(zs), np.array(us)
        out["rw_store_h"], out["rw_log_post"] = M.store["h"], M.store["log_post"]
        out["rw_accept"] = np.array([smp.accept_rate.count["accept"], smp.accept_rate.count["proposal"]], dtype=float)
        zs.clear(), us.clear()
        mdl, st = fresh()
        M = MCMC(st, [NormalNormal("h", model=mdl)], model=mdl, n_burn=0, n_iter=300)
        M.run_mcmc()
        out["nn_z"] = np.concatenate(zs)
        out["nn_store_h"], out["nn_log_post"] = M.store["h"], M.store["log_post"]
    finally:
        stats.norm.rvs, stats.uniform.rvs = saved
    np.savez_compressed(os.path.join(OUT, "example2.npz"), **out)


# ----------------------------------------------------------------------------- truncated conditional
def gen_truncated_conditional():
    """gmrf.gibbs_canonical_truncated_normal (gmrf.py:201-266) called directly on tridiagonal (sparse) and dense
    precisions, and NormalNormal.sample with a truncated prior (sampler.py:199-205) inside the example-4 model
    for a few sweeps; the uniforms behind truncnorm.rvs are recorded."""
    from openmcmc.parameter import LinearCombination, ScaledMatrix

    out = {}
    rng = np.random.default_rng(91)
    used = []

    def _trunc(a, b, loc=0, scale=1, size=None, **_):
        u = rng.random(size)
        used.append(np.asarray(u, dtype=float).reshape(-1))
        return stats.truncnorm.ppf(u, a, b) * scale + loc

    saved = stats.truncnorm.rvs
    stats.truncnorm.rvs = _trunc
    try:
        cases = []
        for n in (1, 2, 9, 40):
            P = sparse.csc_matrix(gmrf.precision_irregular(np.arange(float(n)))) if n > 1 else sparse.csc_matrix(np.array([[1.0]]))
            Q = (3.0 * P + 0.7 * sparse.identity(n, format="csc")).tocsc()
            for lower, upper in ((0.5, 0.7), (-np.inf, 0.2), (-0.3, np.inf), (-4.0, 5.0)):
                cases.append(("tri", n, Q, lower, upper))
        for p_ in (1, 5, 12):
            A = rng.standard_normal((p_, 2 * p_ + 2))
            Q = A @ A.T / (2 * p_ + 2) + 0.5 * np.eye(p_)
            for lower, upper in ((0.5, 0.7), (-1.0, np.inf)):
                cases.append(("dense", p_, (Q + Q.T) / 2, lower, upper))
        out["n_cases"] = len(cases)
        for ci, (kind, n, Q, lower, upper) in enumerate(cases):
            b = rng.standard_normal((n, 1)) * 2
            x0 = np.full((n, 1), 0.6) if (lower, upper) == (0.5, 0.7) else rng.uniform(max(lower, -1), min(upper, 1), size=(n, 1))
            used.clear()
            x = gmrf.gibbs_canonical_truncated_normal(b=b.copy(), Q=Q, x=x0.copy(), lower=lower, upper=upper)
            k = f"c{ci}_"
            Qd = Q.toarray() if sparse.issparse(Q) else Q
            out[k + "kind"], out[k + "n"], out[k + "Q"], out[k + "b"], out[k + "x0"] = kind, n, Qd, b.ravel(), x0.ravel()
            out[k + "lower"], out[k + "upper"] = lower, upper
            out[k + "u"], out[k + "x"] = np.concatenate(used), np.asarray(x).ravel()
        # NormalNormal with a truncated prior inside the example-4 model (sparse route)
        n, n_sweeps = 30, 8
        t = np.arange(n) * 60.0 / n
        y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + np.random.default_rng(3).standard_normal(n)
        P = sparse.csc_matrix(gmrf.precision_irregular(np.arange(float(n)))).tolil()
        P[0, 0] += 1e-3
        mdl = Model([
            Normal("y", mean=LinearCombination(form={"b": "A"}), precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
            Normal("b", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda"), domain_response_lower=np.array(1.5)),
            Gamma("lambda", shape="a_lam", rate="b_lam"), Gamma("tau", shape="a_tau", rate="b_tau")])
        st = {"y": y.copy(), "b": np.maximum(y, 2.0), "mu": np.zeros(n), "lambda": 100, "P_lambda": P.tocsc(), "a_lam": 10,
              "b_lam": 1, "tau": 1, "P_tau": sparse.csc_matrix(np.eye(n)), "a_tau": 1, "b_tau": 1,
              "A": sparse.identity(n, format="csc")}
        gs = []

        def _gamma(a, loc=0, scale=1, size=None, **_):
            g = rng.standard_gamma(np.asarray(a, dtype=np.float64), size=size)
            gs.append(float(np.asarray(g).reshape(-1)[0]))
            return loc + g * scale

        saved_g = stats.gamma.rvs
        stats.gamma.rvs = _gamma
        used.clear()
        try:
            M = MCMC(st, [NormalNormal("b", mdl), NormalGamma("lambda", mdl), NormalGamma("tau", mdl)], model=mdl, n_burn=0,
                     n_iter=n_sweeps)
            M.run_mcmc()
        finally:
            stats.gamma.rvs = saved_g
        Pd = P.toarray()
        out["mc_n"], out["mc_sweeps"], out["mc_y"], out["mc_b0"], out["mc_lower"] = n, n_sweeps, y, np.maximum(y, 2.0), 1.5
        out["mc_P_diag"], out["mc_P_off"] = np.diag(Pd).copy(), np.diag(Pd, -1).copy()
        out["mc_u"] = np.concatenate(used).reshape(n_sweeps, n)
        out["mc_g"] = np.array(gs).reshape(n_sweeps, 2)
        for key in ("b", "lambda", "tau", "log_post"):
            out["mc_store_" + key] = np.asarray(M.store[key])
    finally:
        stats.truncnorm.rvs = saved
    np.savez_compressed(os.path.join(OUT, "truncated_conditional.npz"), **out)


# ----------------------------------------------------------------------------- banded precision (RW2)
def gen_band_chain():
    """Full MCMC.run_mcmc of the example-4 model with a SECOND-order random-walk prior (pentadiagonal precision
    D2'D2 + 1e-3 I, sparse route): NormalNormal + both NormalGamma updates + log_post, recorded draws."""
    from openmcmc.parameter import LinearCombination, ScaledMatrix

    n, n_burn, n_iter = 45, 2, 10
    t = np.arange(n) * 60.0 / n
    y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + 0.3 * np.random.default_rng(8).standard_normal(n)
    D = sparse.diags([np.ones(n - 2), -2 * np.ones(n - 2), np.ones(n - 2)], offsets=[0, 1, 2], shape=(n - 2, n))
    P = (D.T @ D + 1e-3 * sparse.identity(n)).tocsc()
    mdl = Model([
        Normal("y", mean=LinearCombination(form={"b": "A"}), precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
        Normal("b", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
        Gamma("lambda", shape="a_lam", rate="b_lam"), Gamma("tau", shape="a_tau", rate="b_tau")])
    st = {"y": y.copy(), "b": y.copy(), "mu": np.full(n, 0.3), "lambda": 50, "P_lambda": P, "a_lam": 10, "b_lam": 1, "tau": 1,
          "P_tau": sparse.csc_matrix(np.eye(n)), "a_tau": 1, "b_tau": 1, "A": sparse.identity(n, format="csc")}
    rng = np.random.default_rng(17)
    zs, gs = [], []

    def _norm(loc=0, scale=1, size=None, **_):
        z = rng.standard_normal(size)
        zs.append(np.asarray(z, dtype=float).reshape(-1))
        return loc + z * scale

    def _gamma(a, loc=0, scale=1, size=None, **_):
        g = rng.standard_gamma(np.asarray(a, dtype=np.float64), size=size)
        gs.append(float(np.asarray(g).reshape(-1)[0]))
        return loc + g * scale

    saved = (stats.norm.rvs, stats.gamma.rvs)
    stats.norm.rvs, stats.gamma.rvs = _norm, _gamma
    try:
        M = MCMC(st, [NormalNormal("b", mdl), NormalGamma("lambda", mdl), NormalGamma("tau", mdl)], model=mdl, n_burn=n_burn,
                 n_iter=n_iter)
        M.run_mcmc()
    finally:
        stats.norm.rvs, stats.gamma.rvs = saved
    out = {"n": n, "n_burn": n_burn, "n_iter": n_iter, "y": y, "P": P.toarray(), "mu": 0.3,
           "z": np.concatenate(zs).reshape(n_burn + n_iter, n), "g": np.array(gs).reshape(n_burn + n_iter, 2)}
    for key in ("b", "lambda", "tau", "log_post"):
        out["store_" + key] = np.asarray(M.store[key])
    np.savez_compressed(os.path.join(OUT, "band_chain.npz"), **out)


# ----------------------------------------------------------------------------- the reference's own RJ test model
def gen_rj_prior_chain():
    """The model of the reference's reversible-jump unit tests (tests/test_reversible_jump.py fixtures: null
    likelihood, mixture-Normal coefficients, Poisson number of knots, Uniform knot locations, Gamma kernel widths)
    with its sampler list [ManifoldMALA(beta), RandomWalkLoop(theta), RandomWalkLoop(omega), ReversibleJump(n_basis;
    theta, omega; matched beta)], 120 sweeps for three chains, every draw recorded per call site."""
    from openmcmc.distribution.location_scale import NullDistribution
    from openmcmc.sampler.metropolis_hastings import ManifoldMALA

    n_data, n_max, n_iter = 40, 6, 120
    rng0 = np.random.default_rng(5)
    X = np.sort(rng0.uniform(-10, 10, size=(n_data, 1)), axis=0)

    def basis(Xl, knots, scales):
        B = np.full((Xl.shape[0], knots.shape[1]), np.nan)
        for k in range(knots.shape[1]):
            B[:, [k]] = stats.norm.pdf(Xl, loc=knots[:, k], scale=scales[:, k])
        return B

    def move_fn(state, col):
        state["B"] = basis(state["X"], state["theta"], state["omega"])
        return state, 0.0, 0.0

    def birth_fn(cur, prop):
        prop["B"] = basis(prop["X"], prop["theta"], prop["omega"])
        prop["alloc_beta"] = np.concatenate((prop["alloc_beta"], np.array([0], ndmin=2)), axis=0)
        return prop, 0.0, 0.0

    def death_fn(cur, prop, idx):
        prop["B"] = np.delete(prop["B"], obj=idx, axis=1)
        prop["alloc_beta"] = np.delete(prop["alloc_beta"], obj=idx, axis=0)
        return prop, 0.0, 0.0

    mdl = Model([
        NullDistribution(response="y", mean=parameter.LinearCombination(form={"beta": "B"}),
                         precision=parameter.ScaledMatrix(matrix="P", scalar="tau_y")),
        Normal(response="beta", mean=parameter.MixtureParameterVector(param="mu_beta", allocation="alloc_beta"),
               precision=parameter.MixtureParameterMatrix(param="tau_beta", allocation="alloc_beta")),
        Poisson(response="n_basis", rate="rho"),
        Uniform(response="theta", domain_response_lower=np.array([-10.0], ndmin=2), domain_response_upper=np.array([10.0], ndmin=2)),
        Gamma("omega", shape="a_omega", rate="b_omega"),
    ])
    mdl.response = {"y": "mean"}
    K = n_max
    fields = {"mala_z": K, "mala_u": 0, "rwt_u": K, "rwt_acc": K, "rwo_u": K, "rwo_acc": K, "rj_move_u": 0, "rj_theta_u": 0,
              "rj_omega_g": 0, "rj_beta_u": 0, "rj_idx": 0, "rj_acc_u": 0}
    out = {"n_data": n_data, "n_max": n_max, "n_iter": n_iter, "X": X.ravel(), "rho": 4.0, "tau_beta": 0.25, "a_omega": 3.0, "b_omega": 2.0}
    per_chain = []
    for c, k0 in enumerate((4, 1, 6)):
        rng = np.random.default_rng(7100 + c)
        rows, ctx = [], {"where": None, "knot": None, "stage": None}

        def cur():
            return rows[-1]

        def _norm(loc=0, scale=1, size=None, **_):
            z = rng.standard_normal(size)
            flat = np.asarray(z, dtype=float).reshape(-1)
            cur()["mala_z"][: flat.size] = flat
            return loc + z * scale

        def _uniform(loc=0, scale=1, size=None, **_):
            u = rng.random(size)
            w = ctx["where"]
            if w == "beta":
                cur()["mala_u"] = float(u)
            elif w == "theta":
                cur()["rwt_acc"][ctx["knot"]] = float(u)
            elif w == "omega":
                cur()["rwo_acc"][ctx["knot"]] = float(u)
            elif size is not None:
                cur()["rj_theta_u"] = float(np.asarray(u).reshape(-1)[0])
            elif ctx["stage"] == "move":
                cur()["rj_move_u"] = float(u)
            else:
                cur()["rj_acc_u"] = float(u)
            return loc + u * scale

        def _trunc(a, b, loc=0, scale=1, size=None, **_):
            u = rng.random(size)
            v = float(np.asarray(u).reshape(-1)[0])
            w = ctx["where"]
            if w == "theta":
                cur()["rwt_u"][ctx["knot"]] = v
            elif w == "omega":
                cur()["rwo_u"][ctx["knot"]] = v
            else:
                cur()["rj_beta_u"] = v
            return stats.truncnorm.ppf(u, a, b) * scale + loc

        def _gamma(a, loc=0, scale=1, size=None, **_):
            g = rng.standard_gamma(np.asarray(a, dtype=np.float64), size=size)
            cur()["rj_omega_g"] = float(np.asarray(g).reshape(-1)[0])
            return loc + g * scale

        def _randint(low, high, size=None, **_):
            v = int(rng.integers(low, int(np.asarray(high).item())))
            cur()["rj_idx"] = float(v)
            return v

        theta0 = rng.uniform(-10, 10, size=(1, k0))
        omega0 = rng.uniform(0.6, 1.8, size=(1, k0))
        st = {"y": np.zeros((n_data, 1)), "beta": rng.standard_normal((k0, 1)), "tau_y": 100.0, "P": sparse.eye(n_data),
              "B": basis(X, theta0, omega0), "n_basis": k0, "X": X, "theta": theta0, "omega": omega0, "mu_beta": np.zeros((1, 1)),
              "tau_beta": 0.25 * np.ones((1, 1)), "rho": 4.0, "alloc_beta": np.zeros((k0, 1), dtype=int),
              "a_omega": 3.0 * np.ones((1, 1)), "b_omega": 2.0 * np.ones((1, 1))}
        init = {"theta": np.full(K, np.nan), "omega": np.full(K, np.nan), "beta": np.full(K, np.nan)}
        init["theta"][:k0], init["omega"][:k0], init["beta"][:k0] = theta0.ravel(), omega0.ravel(), st["beta"].ravel()
        samplers = [
            ManifoldMALA(param="beta", model=mdl, step=np.array(0.5), max_variable_size=n_max),
            RandomWalkLoop(param="theta", model=mdl, step=np.array(0.1), max_variable_size=n_max,
                           domain_limits=np.array([[-10.0, 10.0]]), state_update_function=move_fn),
            RandomWalkLoop(param="omega", model=mdl, step=np.array(0.1), max_variable_size=n_max,
                           domain_limits=np.array([[0.5, 2.0]]), state_update_function=move_fn),
            ReversibleJump(param="n_basis", model=mdl, associated_params=["theta", "omega"], n_max=n_max,
                           state_birth_function=birth_fn, state_death_function=death_fn,
                           matching_params={"variable": "beta", "matrix": "B", "scale": 1.0, "limits": [-10.0, 10.0]}),
        ]
        names = ["beta", "theta", "omega", "n_basis"]

        def wrap(smp, name):
            inner = smp.sample

            def sample(state):
                if name == "beta":
                    rows.append({k: (np.full(v, np.nan) if v else np.nan) for k, v in fields.items()})
                    rows[-1]["rj_idx"] = -1.0
                ctx["where"], ctx["stage"] = name, "move"
                return inner(state)

            smp.sample = sample

        for smp, name in zip(samplers, names):
            wrap(smp, name)
        for smp in samplers[1:3]:
            def make(inner):
                def prop(state, param_index=None):
                    ctx["knot"] = param_index
                    return inner(state, param_index)
                return prop
            smp.proposal = make(smp.proposal)
        rj_inner = samplers[3].proposal

        def rj_prop(state, param_index=None):
            res = rj_inner(state)
            ctx["stage"] = "accept"
            return res

        samplers[3].proposal = rj_prop
        saved = (stats.norm.rvs, stats.gamma.rvs, stats.uniform.rvs, stats.truncnorm.rvs, stats.randint.rvs)
        stats.norm.rvs, stats.gamma.rvs, stats.uniform.rvs = _norm, _gamma, _uniform
        stats.truncnorm.rvs, stats.randint.rvs = _trunc, _randint
        try:
            M = MCMC(state=st, samplers=samplers, model=mdl, n_burn=0, n_iter=n_iter)
            M.run_mcmc()
        finally:
            stats.norm.rvs, stats.gamma.rvs, stats.uniform.rvs, stats.truncnorm.rvs, stats.randint.rvs = saved
        rec = {"init_theta": init["theta"], "init_omega": init["omega"], "init_beta": init["beta"], "init_k": float(k0)}
        for key in fields:
            rec["tape_" + key] = np.array([row[key] for row in rows])
        for key in ("beta", "theta", "omega", "n_basis", "log_post"):
            rec["store_" + key] = np.asarray(M.store[key])
        for i, nm in enumerate(("mala", "rwt", "rwo", "rj")):
            cnt = samplers[i].accept_rate.count
            rec["accept_" + nm] = np.array([cnt["accept"], cnt["proposal"]], dtype=float)
        per_chain.append(rec)
        print("chain", c, "k0", k0, "visited", np.unique(M.store["n_basis"]), [s.accept_rate.get_acceptance_rate() for s in samplers])
    for key in per_chain[0]:
        out[key] = np.stack([rec[key] for rec in per_chain])
    np.savez_compressed(os.path.join(OUT, "rj_prior_chain.npz"), **out)


# ----------------------------------------------------------------------------- mixture prior
def gen_mixture_chain():
    """The mixture-prior model of the reference's sampler tests (tests/test_sampler.py fixtures: regression response,
    parameter vector with MixtureParameterVector / MixtureParameterMatrix prior, Gamma prior on the per-component
    precisions, Categorical allocation) with samplers [NormalNormal(parameter), NormalGamma(prior_precision_vector),
    MixtureAllocation(allocation)], 40 sweeps, recorded draws."""
    from openmcmc.distribution.distribution import Categorical
    from openmcmc.sampler.sampler import MixtureAllocation

    rng = np.random.default_rng(4)
    n, p_, K, n_iter = 40, 7, 3, 40
    X = rng.standard_normal((n, p_))
    st = {"response": rng.standard_normal((n, 1)), "prefactor_matrix": X, "parameter": rng.standard_normal((p_, 1)),
          "prior_mean": np.array([[-1.0], [0.5], [2.0]]), "precision_matrix": np.diag(rng.random(n) + 0.5),
          "prior_precision_vector": 0.5 + rng.random(K), "gamma_shape": 2.0 * np.ones((K,)), "gamma_rate": 1.0 * np.ones((K,)),
          "allocation": rng.integers(0, K, size=(p_, 1)), "prior_allocation_prob": np.array([[0.2, 0.5, 0.3]])}
    out = {"n": n, "p": p_, "K": K, "n_iter": n_iter, "X": X, "y": st["response"].ravel(), "w": np.diag(st["precision_matrix"]).copy(),
           "parameter0": st["parameter"].ravel(), "prior_mean": st["prior_mean"].ravel(), "prec0": np.asarray(st["prior_precision_vector"]).ravel(),
           "alloc0": st["allocation"].ravel().astype(float), "prob": st["prior_allocation_prob"]}
    mdl = Model([
        Normal("response", mean=parameter.LinearCombination({"parameter": "prefactor_matrix"}), precision=parameter.Identity("precision_matrix")),
        Normal("parameter", mean=parameter.MixtureParameterVector("prior_mean", "allocation"),
               precision=parameter.MixtureParameterMatrix("prior_precision_vector", "allocation")),
        Gamma("prior_precision_vector", shape=parameter.Identity("gamma_shape"), rate=parameter.Identity("gamma_rate")),
        Categorical("allocation", prob="prior_allocation_prob")])
    samplers = [NormalNormal("parameter", mdl), NormalGamma("prior_precision_vector", mdl),
                MixtureAllocation("allocation", mdl, response_param="parameter")]
    rd = np.random.default_rng(61)
    zs, gs, us = [], [], []

    def _norm(loc=0, scale=1, size=None, **_):
        z = rd.standard_normal(size)
        zs.append(np.asarray(z, dtype=float).reshape(-1))
        return loc + z * scale

    def _gamma(a, loc=0, scale=1, size=None, **_):
        g = rd.standard_gamma(np.asarray(a, dtype=np.float64), size=size)
        gs.append(np.asarray(g, dtype=float).reshape(-1))
        return loc + g * scale

    def _uniform(loc=0, scale=1, size=None, **_):
        u = rd.random(size)
        us.append(np.asarray(u, dtype=float).reshape(-1))
        return loc + u * scale

    saved = (stats.norm.rvs, stats.gamma.rvs, stats.uniform.rvs)
    stats.norm.rvs, stats.gamma.rvs, stats.uniform.rvs = _norm, _gamma, _uniform
    try:
        M = MCMC(st, samplers, model=mdl, n_burn=0, n_iter=n_iter)
        M.run_mcmc()
    finally:
        stats.norm.rvs, stats.gamma.rvs, stats.uniform.rvs = saved
    out["z"], out["g"], out["u"] = np.array(zs), np.array(gs), np.array(us)
    for key in ("parameter", "prior_precision_vector", "allocation", "log_post"):
        out["store_" + key] = np.asarray(M.store[key], dtype=float)
    print("allocation counts", np.bincount(M.store["allocation"].astype(int).ravel(), minlength=K))
    np.savez_compressed(os.path.join(OUT, "mixture_chain.npz"), **out)


# ----------------------------------------------------------------------------- gradients
def gen_gradients():
    """Normal.grad_log_p (location_scale.py:190-250) in its three branches and the finite-difference default of the
    Distribution base class (distribution.py:90-198) on a small regression model, for three states."""
    rng = np.random.default_rng(23)
    n, p_ = 12, 4
    X = rng.standard_normal((n, p_))
    w = rng.random(n) + 0.5
    y = rng.standard_normal((n, 1))
    lik = Normal("y", mean=parameter.LinearCombination({"beta": "X"}), precision=parameter.ScaledMatrix("P_tau", "tau"))
    prior_tau = Gamma("tau", shape="a", rate="b")
    out = {"X": X, "w": w, "y": y.ravel(), "a": 2.5, "b": 1.5}
    betas, taus = rng.standard_normal((3, p_)), rng.random(3) + 0.5
    out["beta"], out["tau"] = betas, taus
    for c in range(3):
        st = {"y": y, "X": X, "beta": betas[c].reshape(p_, 1), "P_tau": sparse.diags(w, format="csc"), "tau": np.array([[taus[c]]]),
              "a": np.array([[2.5]]), "b": np.array([[1.5]])}
        g, h = lik.grad_log_p(st, "beta")                       # branch (ii): parameter in the mean
        out[f"c{c}_grad_beta"], out[f"c{c}_hess_beta"] = np.asarray(g).ravel(), np.asarray(h)
        g, h = lik.grad_log_p(st, "tau")                        # branch (iii): finite differences
        out[f"c{c}_grad_tau_lik"], out[f"c{c}_hess_tau_lik"] = float(np.squeeze(g)), float(np.squeeze(h))
        g, h = prior_tau.grad_log_p(st, "tau")                  # base-class default
        out[f"c{c}_grad_tau_prior"], out[f"c{c}_hess_tau_prior"] = float(np.squeeze(g)), float(np.squeeze(h))
        g = lik.grad_log_p(st, "y", hessian_required=False)     # branch (i): the response
        out[f"c{c}_grad_y"] = np.asarray(g).ravel()
    np.savez_compressed(os.path.join(OUT, "gradients.npz"), **out)


# ----------------------------------------------------------------------------- mMALA on regression coefficients
def gen_mala_regression():
    """ManifoldMALA (metropolis_hastings.py:301-373) on the coefficients of a regression: likelihood through the mean
    (grad_log_p branch ii), (a) a ScaledMatrix Gaussian prior, (b) a mixture prior; 40 steps each, recorded draws."""
    from openmcmc.sampler.metropolis_hastings import ManifoldMALA

    rng = np.random.default_rng(29)
    n, p_ = 30, 6
    X = rng.standard_normal((n, p_))
    w = rng.random(n) + 0.5
    y = X @ rng.standard_normal((p_, 1)) + 0.3 * rng.standard_normal((n, 1))
    A = rng.standard_normal((p_, 2 * p_))
    Pm = A @ A.T / (2 * p_) + 0.5 * np.eye(p_)
    Pm = (Pm + Pm.T) / 2
    out = {"X": X, "w": w, "y": y.ravel(), "P": Pm, "tau": 2.0, "lam": 0.7, "mu": 0.3, "step": 1.25, "n_steps": 40,
           "beta0": rng.standard_normal(p_), "prior_mean": np.array([-1.0, 0.5, 2.0]), "prior_prec": np.array([0.4, 1.5, 3.0]),
           "alloc": rng.integers(0, 3, size=p_).astype(float)}
    lik = Normal("y", mean=parameter.LinearCombination({"beta": "X"}), precision=parameter.ScaledMatrix("P_tau", "tau"))
    priors = {"scaled": Normal("beta", mean="mu", precision=parameter.ScaledMatrix("P_lam", "lam")),
              "mixture": Normal("beta", mean=parameter.MixtureParameterVector("prior_mean", "alloc"),
                                precision=parameter.MixtureParameterMatrix("prior_prec", "alloc"))}
    for tag, prior in priors.items():
        mdl = Model([lik, prior])
        st = {"y": y, "X": X, "beta": out["beta0"].reshape(p_, 1).copy(), "P_tau": sparse.diags(w, format="csc"),
              "tau": np.array([[2.0]]), "P_lam": Pm, "lam": np.array([[0.7]]), "mu": np.full((p_, 1), 0.3),
              "prior_mean": out["prior_mean"].reshape(3, 1), "prior_prec": out["prior_prec"].reshape(3, 1),
              "alloc": out["alloc"].astype(int).reshape(p_, 1)}
        smp = ManifoldMALA("beta", mdl, step=np.array(1.25))
        rd = np.random.default_rng(300)
        zs, us, xs, acc = [], [], [], []

        def _norm(loc=0, scale=1, size=None, **_):
            z = rd.standard_normal(size)
            zs.append(np.asarray(z, dtype=float).reshape(-1))
            return loc + z * scale

        def _uniform(loc=0, scale=1, size=None, **_):
            u = rd.random(size)
            us.append(float(u))
            return loc + u * scale

        saved = (stats.norm.rvs, stats.uniform.rvs)
        stats.norm.rvs, stats.uniform.rvs = _norm, _uniform
        try:
            for _ in range(40):
                before = smp.accept_rate.count["accept"]
                st = smp.sample(st)
                xs.append(st["beta"].ravel().copy())
                acc.append(smp.accept_rate.count["accept"] - before)
        finally:
            stats.norm.rvs, stats.uniform.rvs = saved
        out[tag + "_z"], out[tag + "_u"], out[tag + "_x"] = np.array(zs), np.array(us), np.array(xs)
        out[tag + "_accept"] = np.array(acc, dtype=float)
        print(tag, "accepted", int(np.sum(acc)), "of 40")
    np.savez_compressed(os.path.join(OUT, "mala_regression.npz"), **out)


GENERATORS = {"mala_regression": gen_mala_regression, "gradients": gen_gradients, "mixture_chain": gen_mixture_chain, "rj_prior_chain": gen_rj_prior_chain, "band_chain": gen_band_chain, "truncnorm": gen_truncnorm, "rj_gmrf_chain": gen_rj_gmrf_chain, "example2": gen_example2,
              "truncated_conditional": gen_truncated_conditional}

if __name__ == "__main__":
    which = sys.argv[1:] or list(GENERATORS)
    for name in which:
        GENERATORS[name]()
        print(name + ".npz", os.path.getsize(os.path.join(OUT, name + ".npz")))
