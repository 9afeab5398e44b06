"""bench.py --config {cfg2,cfg4,cfg5}: the other BASELINE.json configs as bench lines in the headline's format
(one JSON line: value in chain-updates/s, `roofline` against the fp64 peak where SURVEY.md section 8d rates the config,
`cpu_baseline` = the oracle timed on this host).  Single GPU; the headline line (cfg3) stays bench.py's default.

    cfg2  Bayesian linear regression p = 1000, n = 10 000, 256 chains (dense route)       roofline: fp64 MFMA
    cfg4  ManifoldMALA on a 500-dim correlated Gaussian target, 512 chains per GPU           roofline: fp64 MFMA
    cfg5  reversible jump + GMRF, 5000 nodes, n_max = 20, 512 chains per GPU                  launch/latency bound: no roofline
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector = matrix peak


def _host():
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    blas = None
    try:
        from threadpoolctl import threadpool_info

        blas = [{"api": i.get("internal_api"), "threads": i.get("num_threads")} for i in threadpool_info()]
    except Exception:
        pass
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(), "blas_threadpools": blas}


def _timed(torch, fn, steps, warmup, condition_ms=200.0):
    """W warm-up steps, then exactly K steps between synchronisations; a conditioning run first (clock ramp, bench.py)."""
    t_c = time.perf_counter()
    i = 0
    while time.perf_counter() - t_c < condition_ms * 1e-3:
        fn(i)
        i += 1
        if i % 8 == 0:
            torch.cuda.synchronize()
    for _ in range(warmup):
        fn(i)
        i += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn(i)
        i += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def cfg2(args, torch):
    from openmcmc_amd.engine import Engine
    from oracle import sweep_ref

    n, p, C = 10000, 1000, args.chains or 256
    rng = np.random.default_rng(0)
    X = rng.standard_normal((n, p))
    beta = rng.standard_normal(p)
    y = X @ beta + 0.1 * rng.standard_normal(n)
    eng = Engine(C, seed=1)
    dX, dy = eng.to_device(X), eng.to_device(y)
    for _ in range(3):
        Gram = eng.gram(dX)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        Gram = eng.gram(dX)
    e1.record()
    torch.cuda.synchronize()
    gram_ms = e0.elapsed_time(e1) / 10
    Xty = eng.design_rhs(dX, dy)
    lam, tau = eng.full((C,), 0.01), eng.full((C,), 1.0)
    terms = eng.dense_terms([{"mat": None, "scale": lam}, {"mat": Gram, "rhs": Xty, "scale": tau}], p)
    ident = eng.tridiag_terms([{}], p)
    b, fitted = eng.empty(C, p), eng.empty(C, n)
    q_tau, q_lam, lp, zero = eng.empty(C), eng.empty(1, C), eng.empty(C), eng.zeros(1)

    V, ev = eng.dense_spectral_prepare(Gram)  # once per model: Q_c = lambda_c I + tau_c G in G's eigenbasis
    route = {"spectral": True}

    def sweep(it):  # NormalNormal(beta), NormalGamma(tau), NormalGamma(lambda), log_post, fitted values (mcmc.py:99-111)
        if route["spectral"]:   # what NormalNormal.sample does when no draws are injected (sampler/sampler.py)
            eng.dense_spectral_sample(p, terms, 1, V, ev, b, draw_index=3 * it)
        else:                   # one factorisation per chain: the route of replays with injected draws
            eng.dense_sample_canonical(p, terms, b, draw_index=3 * it)
        eng.design_predict(dX, b, fitted)
        eng.weighted_resid_sq(dy, fitted, q_tau)
        eng.normal_gamma_update(1e-3, 1e-3, n, q_tau, tau, draw_index=3 * it + 1)
        eng.tridiag_quadform(p, ident, b, q_lam)
        eng.normal_gamma_update(1e-3, 1e-3, p, q_lam[0], lam, draw_index=3 * it + 2)
        eng.scaled_gauss_logpdf(n, tau, zero, q_tau, lp)
        eng.scaled_gauss_logpdf(p, lam, zero, q_lam[0], lp, accumulate=True)
        eng.gamma_logpdf(tau, 1e-3, 1e-3, lp, accumulate=True)
        eng.gamma_logpdf(lam, 1e-3, 1e-3, lp, accumulate=True)

    dt = _timed(torch, sweep, args.steps, args.warmup, condition_ms=getattr(args, "condition_ms", 200.0))
    eng.check_status()
    route["spectral"] = False
    dt_chol = _timed(torch, sweep, max(5, args.steps // 20), 2, condition_ms=0.0)
    eng.check_status()
    route["spectral"] = True
    flop = p**3 / 3 + 2 * n * p + 4 * p * p   # SURVEY.md section 8d, per chain-update (factorisation route)
    flop_spec = 4.0 * p * p + 2.0 * n * p + 4 * p   # two p x p products + the fitted values, per chain-update
    achieved = C * flop_spec / dt / 1e12
    out = _line("chain-updates/sec (Bayesian linear regression p=1000, n=10000, 256 chains, 1 GPU)", C / dt, dt, args,
                {"workload": f"BASELINE configs[1]: linreg p={p} n={n}, {C} chains: NormalNormal(beta) through the Gram matrix + "
                             "2x NormalGamma + log_post + fitted values per step (dense route)",
                 "chains_total": C, "check": {"tau_mean": tau.mean().item(), "beta_err": float(np.abs(b.mean(0).cpu().numpy() - beta).max())},
                 "gram_one_off": {"ms": gram_ms, "tflops_on_2p2n": 2.0 * p * p * n / (gram_ms * 1e-3) / 1e12,
                                  "kernel": "k_gram_mfma + k_gram_reduce (own fp64 MFMA kernel)"}})
    out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                       "traffic": None, "alg_flop_per_chain_update": flop_spec,
                       "note": "whole sweep, spectral route: flops actually done (two p x p products per chain + fitted values); "
                               "the section 8d count p^3/3 + 2np + 4p^2 belongs to the factorisation route below",
                       "factorisation_route": {"ms_per_step": 1e3 * dt_chol, "value": C / dt_chol, "alg_flop_per_chain_update": flop,
                                               "achieved": C * flop / dt_chol / 1e12, "frac": C * flop / dt_chol / 1e12 / FP64_PEAK_TFLOPS}}
    if not args.no_cpu:
        k = 2
        rg = np.random.default_rng(1)
        t0 = time.perf_counter()
        sweep_ref.linreg_chain(X, y, 0, k, rg.standard_normal((k, p)), 1 + rg.random((k, 2)))
        per = (time.perf_counter() - t0) / k
        out["cpu_baseline"] = {"value": 1.0 / per, "unit": "chain-updates/s", "cores": "BLAS pool (see blas_threadpools)", "kind": "port",
                               "sample": f"{k} sweeps of 1 chain of the oracle's dense-route restatement (NumPy/LAPACK)", **_host()}
    eng.close()
    return out


def cfg4(args, torch):
    from openmcmc_amd.engine import Engine
    from oracle import mh_ref

    d, C = 500, args.chains or 512
    rng = np.random.default_rng(0)
    A = rng.standard_normal((d, 2 * d))
    Sig = A @ A.T / (2 * d)
    Qh = np.linalg.inv(Sig)
    Qh = (Qh + Qh.T) / 2
    eng = Engine(C, seed=3)
    Q = eng.to_device(Qh)
    step = 0.5
    L, sl = eng.dense_cholesky(Q, 1.0 / step**2)
    x = eng.to_device(np.linalg.solve(np.linalg.cholesky(Qh).T, rng.standard_normal((d, C))).T)
    acc = torch.zeros(C, dtype=torch.int64, device="cuda")
    prop = torch.zeros(C, dtype=torch.int64, device="cuda")

    products = bool(getattr(args, "mala_products", False))
    single = bool(getattr(args, "mala_single", False))
    seen = []

    def one(it):
        if products:   # the step as 3 full + 1 triangular product (omc_mala_step)
            eng.mala_step(Q, None, L, sl, step, x, draw_index=it, accept_count=acc, proposal_count=prop)
        else:          # the step in whitened coordinates (omc_mala_step_white): one launch pair per step
            eng.mala_step_white(None, L, sl, step, x, state_is_current=bool(seen), draw_index=it, accept_count=acc,
                                proposal_count=prop)
            seen.append(1)

    burn_dt = None
    if products or single:
        dt = _timed(torch, one, args.steps, args.warmup, condition_ms=getattr(args, "condition_ms", 200.0))
    else:
        # What ManifoldMALA issues under MCMC.run_mcmc (omc_mala_run_white): blocks of 32 steps per launch, the state of EVERY
        # step and its log density stored (sampler.store + log_post of every iteration, mcmc.py:105-108): one step here is one
        # chain-update of every chain plus that bookkeeping.
        B = 64
        xs, lps = eng.empty(B, C, d), eng.empty(B, C)
        calls = max(1, (args.steps + B - 1) // B)
        it = [0]

        def block(store=True):
            eng.mala_run_white(None, L, sl, step, x, B, state_is_current=it[0] > 0, draw_index0=it[0], draw_stride=1,
                               x_store=xs if store else None, logp_store=lps if store else None, accept_count=acc, proposal_count=prop)
            it[0] += B

        t_c = time.perf_counter()
        while time.perf_counter() - t_c < getattr(args, "condition_ms", 200.0) * 1e-3:
            block()
            torch.cuda.synchronize()
        for _ in range(max(1, args.warmup // B)):
            block()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            block()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (calls * B)
        args.steps = calls * B
        t0 = time.perf_counter()
        for _ in range(calls):
            block(store=False)  # burn-in: nothing stored, no product until the last step
        torch.cuda.synchronize()
        burn_dt = (time.perf_counter() - t0) / (calls * B)
    eng.check_status()
    ref_flop = 14.0 * d * d                              # the reference's algorithm per chain-update
    flop = 9.0 * d * d if products else 2.0 * d * d      # what this route puts on the matrix cores (triangular counted dense)
    achieved = C * flop / dt / 1e12
    out = _line("chain-updates/sec (ManifoldMALA, 500-dim Gaussian target, 512 chains per GPU, 1 GPU)", C / dt, dt, args,
                {"workload": f"BASELINE configs[3]: ManifoldMALA d={d}, step {step}, {C} chains on this GPU (4096 over 8)",
                 "chains_total": C, "check": {"acceptance": acc.sum().item() / max(1, prop.sum().item())}})
    out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                       "traffic": None, "alg_flop_per_chain_update": flop, "reference_flop_per_chain_update": ref_flop,
                       "equivalent_tflops_on_reference_count": C * ref_flop / dt / 1e12,
                       "note": ("3 full + 1 triangular d x d products per chain" if products else
                                "whitened step: ONE triangular d x d product per chain (counted dense, 2 d^2) + an element-wise "
                                "kernel; two launches of ~8 us each -- bound by launch and load latency at this size, not by "
                                "the matrix cores; the reference's algorithm would need 14 d^2")}
    out["config"]["route"] = "omc_mala_step (products)" if products else ("omc_mala_step_white" if single else "omc_mala_run_white")
    if burn_dt is not None:
        out["config"]["burn_in_ms_per_step"] = 1e3 * burn_dt
        out["roofline"]["note"] = ("blocks of 32 whitened steps per launch + ONE triangular d x d by d x (32 C) product per block into the "
                                   "store (counted dense, 2 d^2 per chain-update); every step's state and log density stored; "
                                   "burn_in_ms_per_step: the same steps with nothing stored")
    if not args.no_cpu:
        rg = np.random.default_rng(1)
        xc = np.asarray(np.linalg.solve(np.linalg.cholesky(Qh).T, rg.standard_normal((d, 1))))
        mu0 = np.zeros((d, 1))
        k = 20
        t0 = time.perf_counter()
        for _ in range(k):
            xc, _, _ = mh_ref.mala_step(xc, mu0, Qh, step, rg.standard_normal(d), rg.random())
        per = (time.perf_counter() - t0) / k
        out["cpu_baseline"] = {"value": 1.0 / per, "unit": "chain-updates/s", "cores": "BLAS pool (see blas_threadpools)", "kind": "port",
                               "sample": f"{k} steps of 1 chain of the oracle's restatement (5 factorisations per step, as the reference)", **_host()}
    eng.close()
    return out


def cfg5(args, torch):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from rj_problem import build, make_basis_host

    from openmcmc_amd import gmrf
    from openmcmc_amd.engine import Engine
    from openmcmc_amd.mcmc import MCMC

    n, n_max, C = 5000, 20, args.chains or 512
    rng = np.random.default_rng(0)
    X = np.linspace(-10, 10, n)
    theta_true = np.array([[-6.0, -1.0, 4.5]])
    beta_true = np.array([[3.0], [-2.0], [4.0]])
    b_true = 0.05 * np.cumsum(rng.standard_normal(n)) * np.sqrt(48.0 / n)
    y = (make_basis_host(X.reshape(n, 1), theta_true) @ beta_true).ravel() + b_true + 0.1 * rng.standard_normal(n)
    P = gmrf.precision_irregular(np.arange(float(n))).tolil()
    P[0, 0] += 1e-3
    k0 = np.clip(rng.poisson(5, size=C), 1, n_max)
    init_theta = [rng.uniform(-10, 10, size=k) for k in k0]
    init_beta = [rng.standard_normal(k) for k in k0]
    eng = Engine(C, seed=1)
    mdl, state, samplers = build(y, X, P.tocsc(), n_max, eng, init_theta, init_beta, k0.astype(float))
    import contextlib

    warm = max(args.warmup, 3)
    M = MCMC(state, samplers, model=mdl, n_burn=warm, n_iter=args.steps, n_chains=C, seed=1, engine=eng)
    with contextlib.redirect_stdout(sys.stderr):  # run_mcmc prints the acceptance rates (mcmc.py:113-115)
        M.n_iter = 0                               # the burn-in first, untimed ...
        M.run_mcmc()
        torch.cuda.synchronize()
        M.n_burn, M.n_iter = 0, args.steps         # ... then exactly K stored sweeps
        t0 = time.perf_counter()
        M.run_mcmc()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
    nb = M.store["n_basis"][:, :, 0]
    out = _line("chain-updates/sec (reversible jump + GMRF, 5000 nodes, n_max 20, 512 chains per GPU, 1 GPU)", C / dt, dt, args,
                {"workload": f"BASELINE configs[4]: NormalNormal(b), NormalNormal(beta), 2x NormalGamma, RandomWalkLoop(theta), "
                             f"ReversibleJump(n_basis), store + log_post + fitted values; {C} chains on this GPU",
                 "chains_total": C, "check": {"n_basis_mean": float(nb.mean().item()), "accept_theta": samplers[4].accept_rate.acceptance_rate,
                                              "accept_n_basis": samplers[5].accept_rate.acceptance_rate}})
    out["roofline"] = None  # the RJ index logic is latency / host bound, not roofline-rated
    if not args.no_cpu:
        out["cpu_baseline"] = cfg5_cpu_baseline(y, X, P.tocsc(), n_max, make_basis_host)
    eng.close()
    return out


def cfg5_cpu_baseline(y, X, P, n_max, make_basis_host, seconds_budget=4.0):
    """The oracle's restatement of the whole reversible-jump + GMRF sweep (oracle/rj_sweep_ref.py: the reference's call
    pattern, SuperLU factor + solves for b, matched transitions, full-model log density per knot move) timed on THIS host for
    one chain on a synthetic draw tape; the reference's own rate in the build container is quoted beside it."""
    from oracle import rj_sweep_ref

    n = y.size
    rng = np.random.default_rng(5)

    def tape(S):
        return {"z_b": rng.standard_normal((S, n)), "z_beta": rng.standard_normal((S, n_max)),
                "g": np.stack([rng.standard_gamma(10 + n / 2, size=S), rng.standard_gamma(1 + n / 2, size=S)], axis=1),
                "rw_u": rng.random((S, n_max)), "rw_acc_u": rng.random((S, n_max)), "rj_move_u": rng.random(S),
                "rj_theta_u": rng.random(S), "rj_beta_u": rng.random(S), "rj_acc_u": rng.random(S),
                "rj_idx": rng.integers(0, 1 << 20, size=S).astype(float)}

    model = rj_sweep_ref.RjGmrfModel(y, X, P, make_basis_host, n_max)
    init = {"theta": rng.uniform(-10, 10, size=5), "beta": rng.standard_normal(5)}
    t0 = time.perf_counter()
    rj_sweep_ref.rj_gmrf_chain(model, init, tape(2), 2)
    per = (time.perf_counter() - t0) / 2
    S = int(max(3, min(200, seconds_budget / per)))
    t0 = time.perf_counter()
    rj_sweep_ref.rj_gmrf_chain(model, init, tape(S), S)
    per = (time.perf_counter() - t0) / S
    return {"value": 1.0 / per, "unit": "chain-updates/s", "cores": 1, "kind": "port",
            "sample": f"{S} sweeps of 1 chain (n = {n}, 5 knots at the start) of oracle/rj_sweep_ref.py on this host",
            "reference_in_build_container": {"value": 12.3, "note": "BASELINE.md section 2: the reference itself, 81.5 ms per chain-update, "
                                                                    "8 vCPU Xeon 2.10 GHz"}, **_host()}


def _line(metric, value, dt, args, config):
    return {"metric": metric, "value": value, "unit": "chain-updates/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": config}


def main(args):
    import torch

    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream())
    out = {"cfg2": cfg2, "cfg4": cfg4, "cfg5": cfg5}[args.config](args, torch)
    print(json.dumps(out))
