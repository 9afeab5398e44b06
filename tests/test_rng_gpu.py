"""Device random streams: Philox words bit-exact against the host model (INT path), normal and
gamma draws statistically, and invariance to how chains are sharded over contexts."""

import numpy as np
import pytest
from scipy import stats

import philox_model as pm

pytestmark = pytest.mark.gpu


def make_engine(C, **kw):
    from openmcmc_amd.engine import Engine

    return Engine(C, **kw)


def test_philox_words_bit_exact():
    seed, draw = 0x1234_5678_9ABC_DEF0, (7 << 32) | 99
    eng = make_engine(3, seed=seed, chain_id_offset=(1 << 33) + 5)
    words = eng.fill_philox_u32(1003, draw_index=draw).cpu().numpy().view(np.uint32)
    for c in range(3):
        w = pm.rng_blocks(seed, draw, "raw", (1 << 33) + 5 + c, np.arange(251))
        expect = np.stack(w, axis=1).reshape(-1)[:1003]
        assert np.array_equal(words[c], expect)
    eng.close()


def test_normals_match_host_model_and_are_normal():
    seed = 2024
    eng = make_engine(4, seed=seed, chain_id_offset=10)
    z = eng.fill_normal(20001, draw_index=5).cpu().numpy()
    ref = pm.normals(seed, 5, 12, 20001)
    assert np.max(np.abs(z[2] - ref)) < 1e-12  # same words, libm-level differences only
    flat = z.reshape(-1)
    assert abs(flat.mean()) < 4 / np.sqrt(flat.size)
    assert abs(flat.var() - 1) < 0.02
    assert stats.kstest(flat, "norm").pvalue > 1e-3
    assert abs(np.corrcoef(z[0], z[1])[0, 1]) < 0.03  # chains are independent streams
    assert abs(np.corrcoef(flat[:-1], flat[1:])[0, 1]) < 0.02
    eng.close()


def test_sharding_invariance():
    """Chains 0..7 drawn by one context == chains drawn by two contexts of 4 (global chain id keys
    the stream), for the in-kernel draws of the GMRF sampler and the gamma update."""
    n = 700
    d = np.full(n, 2.0)
    d[0] = d[-1] = 1.0
    d[0] += 1e-3
    off = -np.ones(n - 1)
    y = np.random.default_rng(0).standard_normal(n)

    def run(C, offset, lam_all):
        eng = make_engine(C, seed=77, chain_id_offset=offset)
        lam = eng.to_device(lam_all[offset:offset + C])
        terms = [{"diag": eng.to_device(d), "off": eng.to_device(off), "scale": lam},
                 {"rhs": eng.to_device(y), "center": eng.to_device(y), "scale": eng.full((C,), 1.5)}]
        x, quad, out = eng.empty(C, n), eng.empty(2, C), eng.empty(C)
        eng.tridiag_sample_canonical(n, terms, x, z=None, draw_index=11, quad_out=quad)
        eng.normal_gamma_update(10.0, 1.0, n, quad[0], out, g=None, draw_index=12)
        eng.check_status()
        res = x.cpu().numpy(), out.cpu().numpy()
        eng.close()
        return res

    lam_all = 20 + 10 * np.arange(8.0)
    x8, g8 = run(8, 0, lam_all)
    xa, ga = run(4, 0, lam_all)
    xb, gb = run(4, 4, lam_all)
    assert np.array_equal(x8, np.concatenate([xa, xb]))
    assert np.array_equal(g8, np.concatenate([ga, gb]))


@pytest.mark.parametrize("a0,npos", [(0.3, 0), (1.0, 0), (10.0, 10000), (0.001, 1)])
def test_gamma_draws_distribution(a0, npos):
    C = 20000
    eng = make_engine(C, seed=5)
    quad = eng.full((C,), 3.0)
    out = eng.empty(C)
    eng.normal_gamma_update(a0, 2.0, npos, quad, out, g=None, draw_index=1)
    eng.check_status()
    x = out.cpu().numpy()
    a, b = a0 + npos / 2, 2.0 + 1.5
    assert np.all(x > 0)
    assert stats.kstest(x, "gamma", args=(a, 0, 1 / b)).pvalue > 1e-3
    eng.close()


def test_gamma_zero_rate_guard():
    """b == 0 -> scale = inf (sampler.py:285-286): the draw is +inf, not NaN."""
    eng = make_engine(2)
    out = eng.empty(2)
    eng.normal_gamma_update(1.0, 0.0, 0, eng.zeros(2), out, g=eng.full((2,), 0.7))
    assert np.all(np.isinf(out.cpu().numpy()))
    eng.close()


def test_canonical_draws_have_the_right_law():
    """The reference's statistical check of sample_normal_canonical (tests/test_grmf.py:182-210): with
    in-kernel draws, (x - mu)' Q (x - mu) over many chains is chi^2 with n degrees of freedom, and the
    sample mean converges to Q^-1 b."""
    from scipy import sparse

    from oracle import gmrf_ref

    n, C = 120, 6000
    rng = np.random.default_rng(3)
    d = np.full(n, 2.0) * (1 + 0.2 * rng.random(n))
    d[0] = d[-1] = 1.2
    off = -np.ones(n - 1)
    y = rng.standard_normal(n) + 1
    lam, tau = 30.0, 2.0
    eng = make_engine(C, seed=31)
    terms = [{"diag": eng.to_device(d), "off": eng.to_device(off), "scale": eng.full((C,), lam)},
             {"rhs": eng.to_device(y), "center": eng.to_device(y), "scale": eng.full((C,), tau)}]
    x = eng.empty(C, n)
    eng.tridiag_sample_canonical(n, terms, x, draw_index=2)
    eng.check_status()
    xs = x.cpu().numpy()
    P = sparse.diags((off, d, off), offsets=[-1, 0, 1], format="csc")
    Q = (lam * P + tau * sparse.identity(n)).toarray()
    mu = np.linalg.solve(Q, tau * y)
    r = xs - mu
    maha = np.einsum("ci,ij,cj->c", r, Q, r)
    assert stats.kstest(maha, "chi2", args=(n,)).pvalue > 1e-3
    assert np.max(np.abs(xs.mean(0) - mu)) < 5 * np.sqrt(np.max(np.diag(np.linalg.inv(Q))) / C)
    eng.close()
