"""omc_gmrf_run with several sweeps per launch (workgroup (sweep, chain) = block sweep * C + chain, scales handed from
sweep to sweep through tagged granules): every stored value must equal, bit for bit, what one launch per sweep gives --
same draws (streams are keyed by chain and draw index, not by launch geometry), same arithmetic.
Reference loop: mcmc.py:97-111."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run(n, C, per_launch, n_burn, n_iter, n_thin, generic=False, seed=11, chain_offset=0, C_all=None, reenter=None):
    from openmcmc_amd.engine import Engine

    eng = Engine(C, seed=seed, chain_id_offset=chain_offset)
    eng.set_option("run_sweeps_per_launch", per_launch)
    if reenter is not None:
        eng.set_option("run_reenter", reenter)
    if generic:
        eng.set_option("tridiag_generic", 1)
    rng = np.random.default_rng(0)
    t = np.arange(n) * 60.0 / n
    y = np.sin(t / 20) + 2 * np.cos(t / 12) + 2 + rng.standard_normal(n)
    d = np.full(n, 2.0)
    d[0] = d[-1] = 1.0
    d[0] += 1e-3
    off = -np.ones(n - 1)
    d_y, d_d, d_off = eng.to_device(y), eng.to_device(d), eng.to_device(off)
    C_all = C_all or C  # starting scales belong to the global chain, like the random streams
    lam_all, tau_all = 80.0 + 40 * rng.random(C_all), 0.5 + rng.random(C_all)
    lam, tau = eng.to_device(lam_all[chain_offset:chain_offset + C]), eng.to_device(tau_all[chain_offset:chain_offset + C])
    terms = eng.tridiag_terms([{"diag": d_d, "off": d_off, "scale": lam}, {"rhs": d_y, "center": d_y, "scale": tau}], n)
    logdetP, logdetI = eng.tridiag_logdet(n, d_d, d_off), eng.zeros(1)
    n_slots = n_iter
    nan = float("nan")
    store_b, store_lam, store_tau, store_lp = (eng.full((n_slots, C, n), nan), eng.full((n_slots, C), nan),
                                               eng.full((n_slots, C), nan), eng.full((n_slots, C), nan))
    scratch = eng.empty(C, n)
    blocks = [{"a0": 10.0, "b0": 1.0, "n_pos": n, "store": store_lam, "logdet": logdetP, "draw_index": 1},
              {"a0": 1.0, "b0": 1.0, "n_pos": n, "store": store_tau, "logdet": logdetI, "draw_index": 2}]
    eng.gmrf_run(n, terms, blocks, n_burn, n_iter, n_thin, store_b, scratch, draw_index0=7, draws_per_sweep=3,
                 log_post_store=store_lp)
    eng.check_status()
    assert eng.counter("run_handoff_timeouts") == 0
    out = [t.cpu().numpy() for t in (store_b, store_lam, store_tau, store_lp, lam, tau)]
    eng.close()
    return out


@pytest.mark.parametrize("n,C,generic", [(10000, 300, False), (10000, 3, False), (5000, 40, False), (5000, 40, True),
                                         (2000, 70, False), (700, 5, False)])
def test_sweeps_per_launch_do_not_change_a_bit(n, C, generic):
    ref = run(n, C, 1, 3, 9, 2, generic)
    for per in (16, 5):
        got = run(n, C, per, 3, 9, 2, generic)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b, equal_nan=True)
    assert np.all(np.isfinite(ref[0])) and np.all(ref[1] > 0)


@pytest.mark.parametrize("n,C", [(10000, 300), (10000, 1100), (5000, 40), (9999, 7)])
def test_self_restarting_workgroups_do_not_change_a_bit(n, C):
    """run_reenter: the grid is one workgroup per chain and each restarts itself as its chain's next sweep."""
    ref = run(n, C, 1, 3, 9, 2, reenter=0)
    for per, reenter in ((32, 1), (5, 1), (32, 0), (32, 2), (7, 2)):  # 0: one workgroup per (sweep, chain); 2: no barrier before the restart
        got = run(n, C, per, 3, 9, 2, reenter=reenter)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b, equal_nan=True)


def test_sharding_invariance_of_a_several_sweeps_run():
    full = run(10000, 6, 16, 2, 4, 1)
    a, b = run(10000, 3, 16, 2, 4, 1, C_all=6), run(10000, 3, 16, 2, 4, 1, chain_offset=3, C_all=6)
    for k in range(4):
        assert np.array_equal(full[k][:, :3], a[k]) and np.array_equal(full[k][:, 3:], b[k])
