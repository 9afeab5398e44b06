"""Oracle for the reversible-jump sweep of BASELINE configs[4] / SURVEY.md section 8d cfg5
(reference: sampler/reversible_jump.py, sampler/metropolis_hastings.py:127-289, sampler/sampler.py,
parameter.py:376-538, distribution/distribution.py:377-523, mcmc.py:87-115).
TEST INFRASTRUCTURE ONLY.  One chain, plain numpy, every random draw is an argument.

Model (tests/golden/make_golden_rj.py builds the same one on the reference):
    y     ~ N(B(theta) beta + b, (tau I)^-1)          LinearCombination({"beta": "B", "b": "A"}), A = I
    b     ~ N(0, (lambda P)^-1)                       RW1 precision
    beta  ~ N(mu_beta[alloc], diag(tau_beta[alloc])^-1)   MixtureParameterVector / MixtureParameterMatrix, alloc = 0
    n_basis ~ Poisson(rho),  theta_j ~ U(lo, hi),  lambda ~ Gamma, tau ~ Gamma
Sampler list: NormalNormal(b), NormalNormal(beta), NormalGamma(lambda), NormalGamma(tau),
RandomWalkLoop(theta, truncated, state_update_function = rebuild B), ReversibleJump(n_basis; theta; matched beta).
"""

import numpy as np
from scipy import sparse, stats

from oracle import gmrf_ref, rj_ref, sweep_ref, truncnorm_ref


def matched_birth(B_cur, B_prop, beta, scale, limits, draw):
    """ReversibleJump.matched_birth_transition  [reversible_jump.py:195-261].
    Returns (beta_prop, add to logp_pr_g_cr, add to logp_cr_g_pr)."""
    k1 = B_prop.shape[1]
    G = np.linalg.solve(B_prop.T @ B_prop + 1e-10 * np.eye(k1), B_prop.T @ B_cur)
    F = np.concatenate((G, np.eye(N=k1, M=1, k=-k1 + 1)), axis=1)
    mu_star = G @ beta
    out = mu_star.copy()
    if limits is not None:
        out[-1] = truncnorm_ref.truncated_normal_rv(mu_star[-1], scale, limits[0], limits[1], draw)
        fwd = float(np.squeeze(truncnorm_ref.truncated_normal_log_pdf(out[-1], mu_star[-1], scale, limits[0], limits[1])))
    else:
        out[-1] = mu_star[-1] + scale * draw
        fwd = float(gmrf_ref.gauss_logpdf(out[-1].reshape(1, 1), mu_star[-1].reshape(1, 1), np.array([[1 / scale**2]])))
    with np.errstate(invalid="ignore", divide="ignore"):
        rev = float(np.log(np.linalg.det(F)))
    return out, fwd, rev


def matched_death(B_cur, B_prop, beta, scale, limits, idx):
    """ReversibleJump.matched_death_transition  [reversible_jump.py:263-308]."""
    k = B_cur.shape[1]
    G = np.linalg.solve(B_cur.T @ B_cur + 1e-10 * np.eye(k), B_cur.T @ B_prop)
    F = np.insert(G, obj=idx, values=np.eye(N=k, M=1, k=-idx).flatten(), axis=1)
    mu_aug = np.linalg.solve(F, beta)
    param_del = mu_aug[idx]
    out = np.delete(mu_aug, obj=idx, axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        fwd = float(np.log(np.linalg.det(F)))
    if limits is not None:
        rev = float(np.squeeze(truncnorm_ref.truncated_normal_log_pdf(param_del, 0.0, scale, limits[0], limits[1])))
    else:
        rev = float(gmrf_ref.gauss_logpdf(param_del.reshape(1, 1), np.zeros((1, 1)), np.array([[1 / scale**2]])))
    return out, fwd, rev


class RjGmrfModel:
    """The arithmetic of Model.log_p and of each sampler for the model in the module docstring."""

    def __init__(self, y, X, P, basis_fn, n_max, rho=5.0, mu_beta=0.0, tau_beta=0.25, theta_limits=(-10.0, 10.0),
                 a_lam=10.0, b_lam=1.0, a_tau=1.0, b_tau=1.0, rw_step=0.2, birth_probability=0.5, match_scale=1.0,
                 match_limits=(-10.0, 10.0)):
        self.n = y.size
        self.y = np.asarray(y, dtype=float).reshape(-1, 1)
        self.X = np.asarray(X, dtype=float).reshape(-1, 1)
        self.P = sparse.csc_matrix(P)
        self.I = sparse.identity(self.n, format="csc")
        self.basis_fn, self.n_max, self.rho = basis_fn, n_max, rho
        self.mu_beta, self.tau_beta, self.lim = mu_beta, tau_beta, theta_limits
        self.a_lam, self.b_lam, self.a_tau, self.b_tau = a_lam, b_lam, a_tau, b_tau
        self.rw_step, self.q, self.match_scale, self.match_limits = rw_step, birth_probability, match_scale, match_limits

    # --- Model.log_p  [model.py:57-70]: sum over the seven distributions
    def log_p(self, st):
        k = st["theta"].shape[1]
        fitted = st["B"] @ st["beta"] + st["b"]
        lp = gmrf_ref.gauss_logpdf(self.y, fitted, st["tau"] * self.I)
        lp += gmrf_ref.gauss_logpdf(st["b"], np.zeros((self.n, 1)), st["lambda"] * self.P)
        prec = sparse.diags(np.full(k, self.tau_beta), format="csc")  # parameter.py:501
        lp += gmrf_ref.gauss_logpdf(st["beta"], np.full((k, 1), self.mu_beta), prec)
        lp += float(np.sum(stats.poisson.logpmf(k, self.rho)))      # distribution.py:504-508
        lp += k * -np.log(self.lim[1] - self.lim[0])                # distribution.py:436-442
        lp += sweep_ref.gamma_logpdf(st["lambda"], self.a_lam, self.b_lam)
        lp += sweep_ref.gamma_logpdf(st["tau"], self.a_tau, self.b_tau)
        return float(lp)

    # --- conjugate blocks
    def draw_b(self, st, z):
        like = {"W": st["tau"] * self.I, "y": self.y, "A": self.I, "rest": st["B"] @ st["beta"]}
        x, _, _ = sweep_ref.normal_normal_draw(self.n, (st["lambda"] * self.P, np.zeros((self.n, 1))), [like], z)
        st["b"] = x

    def draw_beta(self, st, z):
        k = st["theta"].shape[1]
        prec = sparse.diags(np.full(k, self.tau_beta), format="csc")
        like = {"W": st["tau"] * self.I, "y": self.y, "A": st["B"], "rest": st["b"]}
        x, _, _ = sweep_ref.normal_normal_draw(k, (prec, np.full((k, 1), self.mu_beta)), [like], z[:k])
        st["beta"] = x

    def draw_lambda(self, st, g):
        a, b = sweep_ref.gamma_conditional(self.a_lam, self.b_lam, st["b"], self.P)
        st["lambda"] = sweep_ref.gamma_draw_from_standard(a, b, g)

    def draw_tau(self, st, g):
        a, b = sweep_ref.gamma_conditional(self.a_tau, self.b_tau, self.y - st["B"] @ st["beta"] - st["b"], self.I)
        st["tau"] = sweep_ref.gamma_draw_from_standard(a, b, g)

    # --- RandomWalkLoop over the knots  [metropolis_hastings.py:276-289, 212-269, 127-173]
    def rw_loop(self, st, u_prop, u_acc, trace=None):
        n_acc = 0
        for j in range(st["theta"].shape[1]):
            prop = dict(st)
            mu = st["theta"][0, j]
            z = float(truncnorm_ref.truncated_normal_rv(mu, self.rw_step, self.lim[0], self.lim[1], u_prop[j]))
            fwd = float(truncnorm_ref.truncated_normal_log_pdf(z, mu, self.rw_step, self.lim[0], self.lim[1]))
            rev = float(truncnorm_ref.truncated_normal_log_pdf(mu, z, self.rw_step, self.lim[0], self.lim[1]))
            prop["theta"] = st["theta"].copy()
            prop["theta"][0, j] = z
            prop["B"] = self.basis_fn(self.X, prop["theta"])  # state_update_function
            log_accept = self.log_p(prop) + rev - (self.log_p(st) + fwd)
            if trace is not None:
                trace["rw_z"][j], trace["rw_lq_fwd"][j], trace["rw_lq_rev"][j] = z, fwd, rev
                trace["rw_log_accept"][j] = log_accept
            if np.log(u_acc[j]) < log_accept:
                st = prop
                n_acc += 1
        return st, n_acc

    # --- ReversibleJump.sample  [reversible_jump.py:76-193, 310-373]
    def rj_step(self, st, u_move, u_theta, draw_beta, idx, u_acc, trace=None):
        k = st["theta"].shape[1]
        birth, _ = rj_ref.move_type(k, self.n_max, self.q, u_move)
        p_birth, p_death = rj_ref.move_probabilities(k, self.n_max, self.q, birth)
        log_prior_theta = -np.log(self.lim[1] - self.lim[0])  # log_p(current, by_observation=True)[-1]
        prop = dict(st)
        if birth:
            new = self.lim[0] + (self.lim[1] - self.lim[0]) * u_theta  # Uniform.rvs  [distribution.py:456-458]
            prop["theta"] = np.concatenate((st["theta"], np.array([[new]])), axis=1)
            prop["B"] = self.basis_fn(self.X, prop["theta"])
            prop["beta"], f, r = matched_birth(st["B"], prop["B"], st["beta"], self.match_scale, self.match_limits, draw_beta)
            fwd = f + np.log(p_birth) + log_prior_theta
            rev = r + np.log(p_death)
        else:
            idx = int(idx) % k  # recorded tapes hold the index itself; synthetic tapes any non-negative integer
            prop["theta"] = np.delete(st["theta"], obj=idx, axis=1)
            prop["B"] = np.delete(st["B"], obj=idx, axis=1)
            prop["beta"], f, r = matched_death(st["B"], prop["B"], st["beta"], self.match_scale, self.match_limits, idx)
            fwd = f + np.log(p_death)
            rev = r + np.log(p_birth) + log_prior_theta
        log_accept = self.log_p(prop) + rev - (self.log_p(st) + fwd)
        if trace is not None:
            trace.update({"rj_birth": float(birth), "rj_idx": -1 if birth else idx, "rj_lq_fwd": fwd, "rj_lq_rev": rev,
                          "rj_log_accept": log_accept,
                          "rj_prop_beta": prop["beta"].ravel().copy(), "rj_prop_theta": prop["theta"].ravel().copy()})
        accepted = bool(np.log(u_acc) < log_accept)
        return (prop if accepted else st), accepted


def rj_gmrf_chain(model, init, tape, n_iter):
    """MCMC.run_mcmc (n_burn = 0) for one chain from the recorded draw tape (see make_golden_rj.Tape).
    Returns (store, traces, accept counters)."""
    n, k_max = model.n, model.n_max
    theta0 = np.asarray(init["theta"], dtype=float).reshape(1, -1)
    st = {"theta": theta0, "B": model.basis_fn(model.X, theta0), "beta": np.asarray(init["beta"], dtype=float).reshape(-1, 1),
          "b": np.zeros((n, 1)), "lambda": float(init.get("lambda", 100.0)), "tau": float(init.get("tau", 10.0))}
    store = {"b": np.full((n, n_iter), np.nan), "beta": np.full((k_max, n_iter), np.nan),
             "theta": np.full((k_max, n_iter), np.nan), "n_basis": np.full((1, n_iter), np.nan),
             "lambda": np.full((1, n_iter), np.nan), "tau": np.full((1, n_iter), np.nan),
             "log_post": np.full((n_iter, 1), np.nan), "y": np.full((n, n_iter), np.nan)}
    traces, acc = [], {"rw": [0, 0], "rj": [0, 0]}
    for it in range(n_iter):
        tr = {"rw_z": np.full(k_max, np.nan), "rw_lq_fwd": np.full(k_max, np.nan), "rw_lq_rev": np.full(k_max, np.nan),
              "rw_log_accept": np.full(k_max, np.nan)}
        model.draw_b(st, tape["z_b"][it])
        model.draw_beta(st, tape["z_beta"][it])
        model.draw_lambda(st, tape["g"][it, 0])
        model.draw_tau(st, tape["g"][it, 1])
        k = st["theta"].shape[1]
        st, n_acc = model.rw_loop(st, tape["rw_u"][it], tape["rw_acc_u"][it], tr)
        acc["rw"][0] += n_acc
        acc["rw"][1] += k
        st, ok = model.rj_step(st, tape["rj_move_u"][it], tape["rj_theta_u"][it], tape["rj_beta_u"][it],
                               tape["rj_idx"][it], tape["rj_acc_u"][it], tr)
        acc["rj"][0] += int(ok)
        acc["rj"][1] += 1
        traces.append(tr)
        k = st["theta"].shape[1]
        store["b"][:, [it]] = st["b"]
        store["beta"][:k, it], store["theta"][:k, it] = st["beta"].ravel(), st["theta"].ravel()
        store["n_basis"][0, it], store["lambda"][0, it], store["tau"][0, it] = k, st["lambda"], st["tau"]
        store["log_post"][it] = model.log_p(st)
        store["y"][:, [it]] = st["B"] @ st["beta"] + st["b"]
    return store, traces, acc
