"""Generic Metropolis-Hastings blocks, ragged state and the matched reversible-jump transitions on
the GPU, each through the C ABI against the reference's golden vectors (tests/golden/truncnorm.npz)
or the pinned CPU oracle (oracle/rj_sweep_ref.py, oracle/truncnorm_ref.py), with injected draws.
fp64 tolerance 1e-10 relative unless a comment says why it is wider."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def make_engine(C, **kw):
    from openmcmc_amd.engine import Engine

    return Engine(C, **kw)


def rel_err(got, ref, floor=1.0):
    return float(np.max(np.abs(got - ref) / np.maximum(floor, np.abs(ref)))) if ref.size else 0.0


def same_infinities(got, ref):
    assert np.array_equal(np.isfinite(got), np.isfinite(ref))
    assert np.array_equal(got[~np.isfinite(ref)], ref[~np.isfinite(ref)])
    return np.isfinite(ref)


# ------------------------------------------------------------------------------------------------ truncated normal
def test_truncated_proposal_matches_reference_grid(golden):
    """omc_rw_propose (truncated) vs gmrf.truncated_normal_rv / truncated_normal_log_pdf of the reference on
    the golden grid: central, one-sided and deep-tail windows, uniforms from 1e-12 to 1 - 1e-9."""
    G = golden("truncnorm")
    mean, scale, lower, upper, u = (G[k] for k in ("mean", "scale", "lower", "upper", "u"))
    combos = sorted({(s, lo, hi) for s, lo, hi in zip(scale, lower, upper)})
    worst = {"x": 0.0, "fwd": 0.0, "rev": 0.0}
    for s, lo, hi in combos:
        sel = (scale == s) & (lower == lo) & (upper == hi)
        C = int(sel.sum())
        eng = make_engine(C)
        x = eng.to_device(mean[sel].reshape(C, 1, 1))
        z = eng.empty(C, 1, 1)
        lqf, lqr = eng.rw_propose(x, z, eng.to_device([s]), eng.to_device([lo]), eng.to_device([hi]),
                                  inject=eng.to_device(u[sel].reshape(C, 1)))
        eng.check_status()
        zg, fg, rg = z.cpu().numpy().ravel(), lqf.cpu().numpy(), lqr.cpu().numpy()
        a, b = (lo - mean[sel]) / s, (hi - mean[sel]) / s
        # SciPy inverts Phi(x) = Phi(a) + u*mass on the LEFT tail whenever a < 0; for Phi(x) -> 1 that loses
        # digits in SciPy itself (its result moves by 1e-8 with 1-ulp changes of the mass), so those grid points
        # are compared at the accuracy SciPy has there, not at 1e-10
        from scipy import special

        xs = (G["x"][sel] - mean[sel]) / s
        ill = (a < 0) & (special.log_ndtr(xs) > -1e-6)
        tol = np.where(ill, 1e-6, RTOL)
        ex = np.abs(zg - G["x"][sel]) / np.maximum(1e-3, np.abs(G["x"][sel]))
        assert np.all(ex <= tol), (s, lo, hi, ex.max())
        worst["x"] = max(worst["x"], float(ex[~ill].max(initial=0.0)))
        # the densities are evaluated at each side's own draw: where the window sits hundreds of sigma from the
        # mean SciPy's ppf is itself ~1e-11 off the exact quantile (checked against mpmath; ours is ~1e-15) and
        # the log-density has slope |x - mean| / scale^2 ~ 1e5 there, so the draw difference is propagated to
        # first order into the bar (slope of the reverse density by finite differences of the oracle)
        from oracle import truncnorm_ref

        dz = np.abs(zg - G["x"][sel])
        slope_f = np.abs(G["x"][sel] - mean[sel]) / s**2
        h = 1e-6 * s
        with np.errstate(invalid="ignore"):
            slope_r = np.abs(truncnorm_ref.truncated_normal_log_pdf(mean[sel], G["x"][sel] + h, s, lo, hi)
                             - truncnorm_ref.truncated_normal_log_pdf(mean[sel], G["x"][sel] - h, s, lo, hi)) / (2 * h)
        slope_r = np.where(np.isfinite(slope_r), slope_r, 0.0)
        ref_f, ref_r = G["logpdf_fwd"][sel], G["logpdf_rev"][sel]
        fin = same_infinities(fg, ref_f)
        ef = np.abs(fg[fin] - ref_f[fin]) / np.maximum(1.0, np.abs(ref_f[fin]))
        bar = np.where(ill[fin], 1e-5, RTOL) + 2 * slope_f[fin] * dz[fin] / np.maximum(1.0, np.abs(ref_f[fin]))
        assert np.all(ef <= bar), (s, lo, hi, ef.max(initial=0.0))
        fin = same_infinities(rg, ref_r)
        er = np.abs(rg[fin] - ref_r[fin]) / np.maximum(1.0, np.abs(ref_r[fin]))
        bar = np.where(ill[fin], 1e-5, RTOL) + 2 * slope_r[fin] * dz[fin] / np.maximum(1.0, np.abs(ref_r[fin]))
        assert np.all(er <= bar), (s, lo, hi, er.max(initial=0.0))
        worst["fwd"], worst["rev"] = max(worst["fwd"], float(ef.max(initial=0.0))), max(worst["rev"], float(er.max(initial=0.0)))
        eng.close()
    print("worst relative errors", worst)


def test_untruncated_proposal_and_gating():
    """Symmetric proposal x + step*z with injected normals, per-element steps, a column of a replicated
    parameter, and the count/index gate (chains whose parameter has fewer columns sit the step out)."""
    import torch

    C, p, n_rep = 37, 3, 5
    rng = np.random.default_rng(3)
    eng = make_engine(C)
    x = rng.standard_normal((C, p, n_rep))
    zin = rng.standard_normal((C, p))
    step = np.array([0.1, 0.5, 2.0])
    count = rng.integers(1, n_rep + 1, size=C).astype(float)
    xd = eng.to_device(x)
    out = xd.clone()
    col = 3
    lqf, lqr = eng.rw_propose(xd, out, eng.to_device(step), column=col, count=eng.to_device(count),
                              inject=eng.to_device(zin))
    eng.check_status()
    exp = x.copy()
    act = col < count
    exp[act, :, col] = x[act, :, col] + step[None, :] * zin[act]
    assert np.array_equal(out.cpu().numpy(), exp)  # one multiply-add: bit-exact
    assert not lqf.cpu().numpy().any() and not lqr.cpu().numpy().any()
    # in-kernel draws: N(0,1) moments over many chains, and a different sub-stream gives different draws
    C2 = 40000
    eng2 = make_engine(C2, seed=11)
    x0 = eng2.zeros(C2, 1, 1)
    z1, z2 = eng2.empty(C2, 1, 1), eng2.empty(C2, 1, 1)
    eng2.rw_propose(x0, z1, eng2.to_device([1.0]), draw_index=3, sub=0)
    eng2.rw_propose(x0, z2, eng2.to_device([1.0]), draw_index=3, sub=4)
    a, b = z1.cpu().numpy().ravel(), z2.cpu().numpy().ravel()
    assert abs(a.mean()) < 0.02 and abs(a.std() - 1) < 0.02 and abs(np.corrcoef(a, b)[0, 1]) < 0.02
    # truncated in-kernel draws stay inside the window and are uniform in probability
    lo, hi = eng2.to_device([-0.5]), eng2.to_device([1.0])
    eng2.rw_propose(x0, z1, eng2.to_device([1.0]), lo, hi, draw_index=9)
    t = z1.cpu().numpy().ravel()
    assert t.min() >= -0.5 and t.max() <= 1.0
    from scipy import stats

    assert stats.kstest(t, stats.truncnorm(-0.5, 1.0).cdf).pvalue > 1e-3
    assert torch.isfinite(z1).all()
    eng.close(), eng2.close()


def test_mh_accept_and_select():
    import torch

    C = 257
    rng = np.random.default_rng(8)
    eng = make_engine(C)
    lp_c, lp_p, f, r = (rng.standard_normal(C) * 3 for _ in range(4))
    u = rng.random(C)
    lp_p[5] = np.nan  # a NaN density rejects (np.log(u) < nan is False)
    count = rng.integers(1, 6, size=C).astype(float)
    index = 2
    acc_n = torch.zeros(C, dtype=torch.int64, device="cuda")
    prop_n = torch.zeros(C, dtype=torch.int64, device="cuda")
    la = eng.empty(C)
    acc = eng.mh_accept(eng.to_device(lp_c), eng.to_device(lp_p), eng.to_device(f), eng.to_device(r),
                        count=eng.to_device(count), index=index, u=eng.to_device(u), accept_count=acc_n,
                        proposal_count=prop_n, log_alpha=la)
    eng.check_status()
    active = index < count
    log_alpha = lp_p + r - (lp_c + f)
    with np.errstate(invalid="ignore"):
        exp = (np.log(u) < log_alpha) & active
    assert np.array_equal(acc.cpu().numpy().astype(bool), exp)
    assert np.array_equal(prop_n.cpu().numpy(), active.astype(np.int64))
    assert np.array_equal(acc_n.cpu().numpy(), exp.astype(np.int64))
    got_la = la.cpu().numpy()
    assert np.array_equal(got_la[active & ~np.isnan(log_alpha)], log_alpha[active & ~np.isnan(log_alpha)])
    # select
    src, dst = rng.standard_normal((C, 7, 3)), rng.standard_normal((C, 7, 3))
    d = eng.to_device(dst)
    eng.chain_select(acc, eng.to_device(src), d)
    assert np.array_equal(d.cpu().numpy(), np.where(exp[:, None, None], src, dst))
    # in-kernel uniforms: acceptance frequency of log_alpha = log(0.3) is 0.3
    C2 = 50000
    eng2 = make_engine(C2, seed=2)
    zero = eng2.zeros(C2)
    acc2 = eng2.mh_accept(zero, eng2.full((C2,), np.log(0.3)), draw_index=1, sub=7)
    assert abs(acc2.cpu().numpy().mean() - 0.3) < 0.01
    eng.close(), eng2.close()


def test_ragged_resize_matches_numpy():
    """np.concatenate / np.delete per chain on padded arrays, both layouts (rows ragged, columns ragged, and a
    column-major basis matrix)."""
    import torch

    C, kmax, rows = 41, 6, 5
    rng = np.random.default_rng(12)
    count = rng.integers(1, kmax + 1, size=C)
    birth = (rng.random(C) < 0.5) & (count < kmax)
    birth |= count == 1
    dele = np.array([-1 if b else rng.integers(0, k) for b, k in zip(birth, count)], dtype=np.int64)
    eng = make_engine(C)
    cd = eng.to_device(count.astype(float))
    bd = torch.as_tensor(birth.astype(np.int32), device="cuda")
    dd = torch.as_tensor(dele, device="cuda")

    def expect(val, new):  # val: (C, rows, kmax) padded
        out = np.zeros_like(val)
        for c in range(C):
            live = val[c][:, : count[c]]
            res = np.concatenate((live, new[c].reshape(-1, 1)), axis=1) if birth[c] else np.delete(live, dele[c], axis=1)
            out[c][:, : res.shape[1]] = res
        return out

    val = rng.standard_normal((C, rows, kmax))
    for c in range(C):
        val[c][:, count[c]:] = 0.0
    new = rng.standard_normal((C, rows))
    exp = expect(val, new)
    got = eng.ragged_resize(eng.to_device(val), cd, bd, dd, axis=1, new_vals=eng.to_device(new))
    assert np.array_equal(got.cpu().numpy(), exp)
    # the same matrix kept column-major per chain (basis layout)
    phys = eng.to_device(np.ascontiguousarray(val.transpose(0, 2, 1)))  # (C, kmax, rows)
    got = eng.ragged_resize(phys.transpose(1, 2), cd, bd, dd, axis=1, new_vals=eng.to_device(new))
    assert got.transpose(1, 2).is_contiguous() and np.array_equal(got.cpu().numpy(), exp)
    # rows ragged: beta (k, 1)
    vec = val[:, 0, :].reshape(C, kmax, 1)
    got = eng.ragged_resize(eng.to_device(vec), cd, bd, dd, axis=0, new_vals=eng.to_device(new[:, :1].copy()))
    assert np.array_equal(got.cpu().numpy()[:, :, 0], expect(val[:, :1, :], new[:, :1])[:, 0, :])
    eng.check_status()
    eng.close()


# ------------------------------------------------------------------------------------------------ per-chain designs
def ragged_problem(C, n, kmax, seed):
    rng = np.random.default_rng(seed)
    count = rng.integers(1, kmax + 1, size=C)
    count[0], count[1] = 1, kmax
    B = rng.standard_normal((C, kmax, n))
    beta = rng.standard_normal((C, kmax))
    for c in range(C):
        B[c, count[c]:] = 0.0
        beta[c, count[c]:] = 0.0
    return rng, count, B, beta


def test_design_predict_and_gram_batched():
    C, n, kmax = 19, 333, 7
    rng, count, B, beta = ragged_problem(C, n, kmax, 5)
    eng = make_engine(C)
    addc, adds, tau = rng.standard_normal((C, n)), rng.standard_normal(n), rng.random(C) + 0.5
    Bd, bd = eng.to_device(B), eng.to_device(beta)
    got = eng.design_predict_batched(Bd, bd, add_chain=eng.to_device(addc), add_shared=eng.to_device(adds), alpha=-1.0,
                                     chain_scale=eng.to_device(tau)).cpu().numpy()
    exp = tau[:, None] * (-np.einsum("cjn,cj->cn", B, beta) + addc + adds[None, :])
    assert rel_err(got, exp) < 1e-13
    got = eng.design_predict_batched(Bd, bd).cpu().numpy()
    assert rel_err(got, np.einsum("cjn,cj->cn", B, beta)) < 1e-13
    w = rng.random(n) + 0.1
    gram, rhs = eng.design_gram_batched(Bd, w=eng.to_device(w), resid_shared=eng.to_device(adds), resid_chain=eng.to_device(addc))
    eng.check_status()
    exp_g = np.einsum("cin,n,cjn->cij", B, w, B)
    exp_r = np.einsum("cin,n,cn->ci", B, w, adds[None, :] - addc)
    assert rel_err(gram.cpu().numpy(), exp_g) < 1e-12 and rel_err(rhs.cpu().numpy(), exp_r) < 1e-12
    gram2, none = eng.design_gram_batched(Bd)
    assert none is None and rel_err(gram2.cpu().numpy(), np.einsum("cin,cjn->cij", B, B)) < 1e-12
    eng.close()


def test_small_ragged_normal_normal_matches_oracle():
    """omc_small_sample_canonical vs the oracle's Rue-Held draw on the live block of every chain."""
    from scipy import sparse

    from oracle import gmrf_ref

    C, n, kmax = 23, 90, 6
    rng, count, B, beta = ragged_problem(C, n, kmax, 9)
    eng = make_engine(C)
    y, b = rng.standard_normal(n), rng.standard_normal((C, n))
    tau = rng.random(C) * 5 + 0.2
    prec = rng.random((C, kmax)) + 0.1
    pmean = rng.standard_normal((C, kmax))
    z = rng.standard_normal((C, kmax))
    gram, rhs = eng.design_gram_batched(eng.to_device(B), resid_shared=eng.to_device(y), resid_chain=eng.to_device(b))
    mu = eng.empty(C, kmax)
    x = eng.small_sample_canonical(gram, rhs, eng.to_device(prec), lik_scale=eng.to_device(tau), prior_mean=eng.to_device(pmean),
                                   count=eng.to_device(count.astype(float)), z=eng.to_device(z), mean_out=mu)
    eng.check_status()
    xg, mg = x.cpu().numpy(), mu.cpu().numpy()
    for c in range(C):
        k = count[c]
        Bc = B[c, :k].T  # (n, k)
        Q = sparse.diags(prec[c, :k]).toarray() + tau[c] * Bc.T @ Bc
        rhs_c = (prec[c, :k] * pmean[c, :k] + tau[c] * Bc.T @ (y - b[c])).reshape(k, 1)
        xo, mo, _ = gmrf_ref.draw_canonical(rhs_c, Q, z[c, :k].reshape(k, 1))
        assert rel_err(xg[c, :k], xo.ravel()) < RTOL and rel_err(mg[c, :k], mo.ravel()) < RTOL
        assert not xg[c, k:].any()
    # a non-positive-definite block is latched like every other factorisation failure
    bad = prec.copy()
    bad[3, 0] = -1e6
    eng.small_sample_canonical(gram, rhs, eng.to_device(bad), lik_scale=eng.to_device(tau), count=eng.to_device(count.astype(float)),
                               z=eng.to_device(z))
    with pytest.raises(np.linalg.LinAlgError, match="chain 3"):
        eng.check_status()
    eng.close()


@pytest.mark.parametrize("limits", [(-10.0, 10.0), None])
def test_matched_transitions_match_oracle(limits):
    """omc_rj_matched_transition vs ReversibleJump.matched_birth/death_transition restated in the oracle, on
    Gaussian-kernel bases with random knots (incl. nearly coincident knots, the ill-conditioned case)."""
    import torch

    from oracle import rj_sweep_ref

    C, n, kmax = 64, 120, 7
    rng = np.random.default_rng(21)
    X = np.linspace(-10, 10, n)
    count = rng.integers(1, kmax + 1, size=C)
    count[:4] = [1, 2, kmax - 1, kmax]
    birth = (rng.random(C) < 0.5)
    birth = np.where(count == 1, True, np.where(count == kmax, False, birth))
    dele = np.array([-1 if b else rng.integers(0, k) for b, k in zip(birth, count)], dtype=np.int64)
    theta = rng.uniform(-10, 10, size=(C, kmax))
    theta[7, 1] = theta[7, 0] + 0.05  # close knots: cond(X'X) ~ 1e3
    new_theta = rng.uniform(-10, 10, size=C)
    beta = rng.standard_normal((C, kmax)) * 2
    draw = rng.random(C) if limits is not None else rng.standard_normal(C)

    def basis(th):
        return np.exp(-0.5 * (X[:, None] - th[None, :]) ** 2) / np.sqrt(2 * np.pi)

    Bc, Bp = np.zeros((C, kmax, n)), np.zeros((C, kmax, n))
    exp_beta, exp_f, exp_r = np.zeros((C, kmax)), np.zeros(C), np.zeros(C)
    for c in range(C):
        k = count[c]
        th = theta[c, :k]
        cur = basis(th)
        Bc[c, :k] = cur.T
        beta[c, k:] = 0.0
        if birth[c]:
            prop = basis(np.append(th, new_theta[c]))
            out, f, r = rj_sweep_ref.matched_birth(cur, prop, beta[c, :k].reshape(k, 1), 1.3, limits, draw[c])
        else:
            prop = np.delete(cur, dele[c], axis=1)
            out, f, r = rj_sweep_ref.matched_death(cur, prop, beta[c, :k].reshape(k, 1), 1.3, limits, int(dele[c]))
        Bp[c, : prop.shape[1]] = prop.T
        exp_beta[c, : out.size], exp_f[c], exp_r[c] = out.ravel(), f, r
    eng = make_engine(C)
    gc, _ = eng.design_gram_batched(eng.to_device(Bc))
    gp, _ = eng.design_gram_batched(eng.to_device(Bp))
    lqf, lqr = eng.full((C,), 0.25), eng.full((C,), -0.5)  # pre-existing contributions are added to
    got = eng.rj_matched_transition(gc, gp, eng.to_device(count.astype(float)), torch.as_tensor(birth.astype(np.int32), device="cuda"),
                                    torch.as_tensor(dele, device="cuda"), eng.to_device(beta), 1.3, limits, lqf, lqr,
                                    inject=eng.to_device(draw))
    eng.check_status()
    gb, gf, gr = got.cpu().numpy(), lqf.cpu().numpy() - 0.25, lqr.cpu().numpy() + 0.5
    # the transition solves with X'X + 1e-10 I: rounding is amplified by its condition number, so the bar is
    # 1e-10 * cond-ish; 1e-8 covers the close-knot chain, everything else sits near 1e-13
    assert rel_err(gb, exp_beta) < 1e-8 and rel_err(gf, exp_f) < 1e-8 and rel_err(gr, exp_r) < 1e-8
    easy = np.arange(C) != 7
    assert rel_err(gb[easy], exp_beta[easy]) < RTOL and rel_err(gf[easy], exp_f[easy]) < RTOL
    assert rel_err(gr[easy], exp_r[easy]) < RTOL
    eng.close()


def test_matched_transition_accuracy_when_ill_conditioned():
    """Two knots 0.00025 apart make X'X + 1e-10 I nearly singular (cond ~ 1e8): the reference's LU solve is then only
    accurate to ~1e-8, so agreement with it cannot be the bar.  Here the kernel is compared with EXACT arithmetic
    (mpmath, 60 digits): it must be at least as accurate as the reference's algorithm (the oracle), up to a factor."""
    import mpmath as mp
    import torch

    from oracle import rj_sweep_ref

    mp.mp.dps = 60
    n, kmax = 48, 6
    X = np.linspace(-10, 10, n)
    theta = np.array([-6.1, -0.7, -0.70025, 3.3, 7.9])
    beta = np.array([2.5, -1.3, 3.1, 0.4, -2.2])
    k, idx = theta.size, 2

    def basis(th):
        return np.exp(-0.5 * (X[:, None] - th[None, :]) ** 2) / np.sqrt(2 * np.pi)

    cur = basis(theta)
    prop = np.delete(cur, idx, axis=1)
    ref_out, ref_f, _ = rj_sweep_ref.matched_death(cur, prop, beta.reshape(k, 1), 1.0, (-10.0, 10.0), idx)
    Bm = mp.matrix(cur.tolist())
    Mm = Bm.T * Bm + mp.mpf("1e-10") * mp.eye(k)
    Rm = Bm.T * mp.matrix(prop.tolist())
    cols = [mp.lu_solve(Mm, Rm[:, j]) for j in range(k - 1)]
    F = mp.zeros(k, k)
    for i in range(k):
        jj = 0
        for j in range(k):
            if j == idx:
                F[i, j] = 1 if i == idx else 0
            else:
                F[i, j] = cols[jj][i]
                jj += 1
    mu = mp.lu_solve(F, mp.matrix(beta.tolist()))
    exact = np.array([float(mu[i]) for i in range(k) if i != idx])
    exact_f = float(mp.log(mp.det(F)))
    C = 2
    eng = make_engine(C)
    Bc, Bp = np.zeros((C, kmax, n)), np.zeros((C, kmax, n))
    Bc[:, :k], Bp[:, : k - 1] = cur.T, prop.T
    coef = np.zeros((C, kmax))
    coef[:, :k] = beta
    gc, _ = eng.design_gram_batched(eng.to_device(Bc))
    gp, _ = eng.design_gram_batched(eng.to_device(Bp))
    lqf, lqr = eng.zeros(C), eng.zeros(C)
    got = eng.rj_matched_transition(gc, gp, eng.full((C,), float(k)), torch.zeros(C, dtype=torch.int32, device="cuda"),
                                    torch.full((C,), idx, dtype=torch.int64, device="cuda"), eng.to_device(coef), 1.0,
                                    (-10.0, 10.0), lqf, lqr)
    eng.check_status()
    g = got.cpu().numpy()[0, : k - 1]
    err_ref = np.max(np.abs(ref_out.ravel() - exact))
    err_gpu = np.max(np.abs(g - exact))
    cond = np.linalg.cond(cur.T @ cur + 1e-10 * np.eye(k))
    print(f"cond {cond:.3g}: |reference - exact| = {err_ref:.3g}, |kernel - exact| = {err_gpu:.3g}")
    assert cond > 1e7 and err_gpu < 4 * err_ref + cond * 1e-17
    assert abs(lqf.cpu().numpy()[0] - exact_f) < 4 * abs(ref_f - exact_f) + cond * 1e-17
    eng.close()


def test_ragged_log_densities():
    from scipy import stats

    C, kmax = 50, 8
    rng = np.random.default_rng(4)
    count = rng.integers(0, kmax + 1, size=C).astype(float)
    x, mean = rng.standard_normal((C, kmax)), rng.standard_normal((C, kmax))
    prec = rng.random((C, kmax)) + 0.2
    eng = make_engine(C)
    out = eng.full((C,), 1.5)
    eng.diag_gauss_logpdf(eng.to_device(x), eng.to_device(prec), out, mean=eng.to_device(mean), count=eng.to_device(count),
                          accumulate=True)
    exp = np.array([stats.norm.logpdf(x[c, : int(k)], mean[c, : int(k)], 1 / np.sqrt(prec[c, : int(k)])).sum()
                    for c, k in enumerate(count)]) + 1.5
    assert rel_err(out.cpu().numpy(), exp) < 1e-13
    eng.poisson_logpmf(eng.to_device(count), 5.0, out)
    assert rel_err(out.cpu().numpy(), stats.poisson.logpmf(count, 5.0)) < 1e-13
    eng.count_logpdf(eng.to_device(count), -np.log(20.0), out, accumulate=True)
    assert rel_err(out.cpu().numpy(), stats.poisson.logpmf(count, 5.0) - count * np.log(20.0)) < 1e-13
    param = np.array([0.25, 4.0, 9.0])
    alloc = rng.integers(0, 3, size=(C, kmax)).astype(float)
    got = eng.mixture_gather(eng.to_device(param), eng.to_device(alloc), count=eng.to_device(count), fill=1.0).cpu().numpy()
    exp = np.where(np.arange(kmax)[None, :] < count[:, None], param[alloc.astype(int)], 1.0)
    assert np.array_equal(got, exp)
    eng.check_status()
    eng.close()


@pytest.mark.parametrize("n", [700, 2500])  # one row block (single launch) and several (parts + ordered sum)
def test_design_resid_sq_matches_predict_then_residual(n):
    """omc_design_resid_sq_batched = the regression quadratic form without the fitted values in memory: against
    omc_design_predict_batched + omc_weighted_resid_sq and against numpy, ragged live columns, optional terms."""
    from openmcmc_amd.engine import Engine

    rng = np.random.default_rng(5)
    C, kmax = 7, 9
    eng = Engine(C, seed=1)
    B = rng.standard_normal((C, kmax, n))
    coef = rng.standard_normal((C, kmax))
    for c in range(C):
        coef[c, rng.integers(1, kmax + 1):] = 0.0  # dead columns carry zero coefficients
    y, off, sh, w = rng.standard_normal(n), rng.standard_normal((C, n)), rng.standard_normal(n), 0.5 + rng.random(n)
    dB, dc, dy, doff, dsh, dw = (eng.to_device(v) for v in (B, coef, y, off, sh, w))
    for add_chain, add_shared, ww in ((None, None, None), (doff, None, dw), (doff, dsh, dw), (None, dsh, None)):
        got = eng.design_resid_sq_batched(dB, dc, dy, add_chain=add_chain, add_shared=add_shared, w=ww).cpu().numpy()
        fitted = eng.design_predict_batched(dB, dc, add_chain=add_chain, add_shared=add_shared)
        two_step = eng.empty(C)
        eng.weighted_resid_sq(dy, fitted, two_step, w=ww)
        f = np.einsum("ckn,ck->cn", B, coef) + (off if add_chain is not None else 0.0) + (sh if add_shared is not None else 0.0)
        ref = (((y - f) ** 2) * (w if ww is not None else 1.0)).sum(axis=1)
        assert np.max(np.abs(got - ref) / ref) < 1e-12
        assert np.max(np.abs(got - two_step.cpu().numpy()) / ref) < 1e-12
    eng.close()
