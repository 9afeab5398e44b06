"""The step directly behind the sampler loop (reference mcmc.py:105-111, sampler/sampler.py:69-118):
the store stays on the device, is reduced there (quantiles next to the moments), travels thinned, and -- as a ring that a second
stream drains while the chains keep sampling -- is no longer bounded by what the GPU holds.

Bars: quantiles bit-equal to np.quantile / np.nanquantile of the collected store; a run with n_iter > ring length returns the
full trace bit-equal to an un-ringed run (fused C loop, sampler-by-sampler loop, variable-size reversible-jump store)."""

import os
import subprocess
import sys

import numpy as np
import pytest

from test_mcmc_api_gpu import build

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QS = [0.0, 0.025, 0.25, 0.5, 0.9, 0.975, 1.0]  # seven levels: two passes of the four-level kernel


def make_engine(C, seed=0):
    from openmcmc_amd.engine import Engine

    return Engine(C, seed=seed)


def same(got, want):
    """bit-for-bit up to the sign of zero and the payload of NaN; says where it is not"""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    bad = ~((got == want) | (np.isnan(got) & np.isnan(want)))
    if bad.any():
        idx = np.argwhere(bad)[:6]
        raise AssertionError("differs at " + "; ".join(f"{tuple(i)}: {got[tuple(i)]!r} vs {want[tuple(i)]!r}" for i in idx))
    return True


def awkward_store(rng, n_iter, C, size):
    """ties, both zeros, infinities, a constant column, values of both signs over many binades"""
    a = rng.standard_normal((n_iter, C, size)) * np.exp(rng.uniform(-30, 30, size=(1, 1, size)))
    a[:, :, 0] = 1.5                                   # a constant element
    a[:, :, 1] = rng.integers(-2, 3, size=(n_iter, C))  # heavy ties, incl. +0.0
    a[::3, :, 1] *= -1.0                                # ... and -0.0
    a[0, 0, 2], a[-1, -1, 2] = np.inf, -np.inf
    a[:, :, 3] = np.abs(a[:, :, 3])                    # one sign only
    return a


@pytest.mark.parametrize("shape", [(1, 3, 5), (2, 1, 1), (37, 5, 23), (300, 64, 1000), (2100, 2, 5000), (3, 2, 9000)])
@pytest.mark.filterwarnings("ignore:invalid value")
def test_store_quantiles_are_numpys(shape):
    """(shapes chosen to reach all three kernels: row-sliced global histograms for few columns; a workgroup per 8 columns with 8-bit
    digits for many long columns (2100 iterations x 10 000 columns per chain); per 16 columns with 4-bit digits for many short ones)"""
    n_iter, C, size = shape
    rng = np.random.default_rng(sum(shape))
    a = awkward_store(rng, n_iter, C, size) if size >= 5 else rng.standard_normal(shape)
    eng = make_engine(C)
    t = eng.to_device(a)
    got = eng.store_quantiles(t, QS, pooled=False, omit_nan=False).cpu().numpy()
    want = np.quantile(a, QS, axis=0)               # (nq, C, size): per chain over its iterations
    # (inf - inf inside numpy's interpolation is NaN there and here: equal_nan)
    assert same(got, want)
    got = eng.store_quantiles(t, QS, pooled=True, omit_nan=False).cpu().numpy()
    want = np.quantile(a.reshape(n_iter * C, size), QS, axis=0)
    assert same(got, want)
    assert same(eng.store_quantiles(t, 0.5, pooled=True).cpu().numpy()[0], np.quantile(a.reshape(-1, size), 0.5, axis=0))  # a scalar level
    eng.check_status()
    eng.close()


def test_store_quantiles_with_the_nan_padding_of_variable_size_entries():
    n_iter, C, size = 41, 4, 9
    rng = np.random.default_rng(3)
    a = rng.standard_normal((n_iter, C, size))
    live = rng.integers(1, size + 1, size=(n_iter, C, 1))
    a[np.arange(size).reshape(1, 1, -1) >= live] = np.nan  # NaN beyond the live length (sampler.py:81-87)
    a[:, 2, size - 1] = np.nan                              # an element one chain never had
    a[:, :, size - 2] = np.nan                              # ... and one no chain ever had
    eng = make_engine(C)
    t = eng.to_device(a)
    with np.errstate(all="ignore"):
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want_c = np.nanquantile(a, QS, axis=0)
            want_p = np.nanquantile(a.reshape(-1, size), QS, axis=0)
            prop_c = np.quantile(a, QS, axis=0)
    got = eng.store_quantiles(t, QS, pooled=False, omit_nan=True).cpu().numpy()
    assert same(got, want_c)
    assert np.all(np.isnan(got[:, 2, size - 1])) and np.all(np.isnan(got[:, :, size - 2]))
    assert same(eng.store_quantiles(t, QS, pooled=True, omit_nan=True).cpu().numpy(), want_p)
    assert same(eng.store_quantiles(t, QS, pooled=False, omit_nan=False).cpu().numpy(), prop_c)
    with pytest.raises(ValueError):
        eng.store_quantiles(t, [0.5, 1.5])
    eng.close()


def test_quantiles_summary_and_thinned_collect_through_mcmc(golden):
    G = golden("gmrf_chain")
    M, _ = build(G, "sparse_", True, 6, fuse=True, n_burn=3, n_iter=25, seed=5)
    M.run_mcmc()
    out = M.collect()
    for key in ("b", "lambda", "log_post"):
        arr = out[key] if key != "log_post" else np.transpose(out[key], (0, 2, 1))  # (C, size, n_iter)
        assert np.array_equal(M.quantiles(key, QS, pooled=False), np.moveaxis(np.quantile(arr, QS, axis=2), 0, 0))
        flat = np.transpose(arr, (1, 0, 2)).reshape(arr.shape[1], -1)               # (size, C n_iter)
        assert np.array_equal(M.quantiles(key, QS, pooled=True), np.quantile(flat, QS, axis=1))
        mean, var = M.summary(key, pooled=True)
        assert np.max(np.abs(mean - flat.mean(axis=1))) <= 1e-12 * np.max(np.abs(flat))
        assert np.max(np.abs(var - flat.var(axis=1, ddof=1))) <= 1e-10 * np.max(flat.var(axis=1, ddof=1))
    thin = M.collect(every=4)
    for key in out:
        want = out[key][..., ::4] if key != "log_post" else out[key][:, ::4]
        assert np.array_equal(thin[key], want), key
    M.engine.close()


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("ring", [2, 6, 7, 64])
def test_ring_store_returns_the_unringed_trace(golden, fuse, ring):
    """n_iter = 23 stored iterations through a ring of `ring` slabs (halves of 1, 3, 3 and -- ring longer than the run -- all 23):
    every stored entry bit-equal to the run that keeps the whole store on the device."""
    G = golden("gmrf_chain")
    M0, _ = build(G, "sparse_", True, 5, fuse=fuse, n_burn=4, n_iter=23, seed=9)
    M0.run_mcmc()
    want = M0.collect()
    M0.engine.close()
    M1, _ = build(G, "sparse_", True, 5, fuse=fuse, n_burn=4, n_iter=23, seed=9, store_ring=ring)
    M1.run_mcmc()
    assert M1.store["b"].shape[0] == (23 if ring >= 23 else 2 * (ring // 2))
    got = M1.collect()
    assert set(got) == set(want)
    for key in want:
        assert got[key].shape == want[key].shape and np.array_equal(got[key], want[key]), key
    if ring < 23:
        with pytest.raises(ValueError):
            M1.summary("b")
    thin = M1.collect(every=5)
    assert np.array_equal(thin["b"], want["b"][..., ::5])
    M1.engine.close()


def test_ring_store_of_a_variable_size_model(golden):
    """The reversible-jump model's stores are NaN beyond the live length (sampler.py:105-116): a reused ring slab must start as
    the reference's fresh store does."""
    from test_rj_chain_gpu import run_with_tape

    G = golden("rj_gmrf_chain")
    chains = np.arange(G["init_k"].shape[0])
    n_iter = 40
    M0, _, _ = run_with_tape(G, chains, n_iter)
    M0.run_mcmc()
    want = M0.collect()
    M0.engine.close()
    M1, _, _ = run_with_tape(G, chains, n_iter, store_ring=8)
    M1.run_mcmc()
    got = M1.collect()
    for key in want:
        assert np.array_equal(got[key], want[key], equal_nan=True), key
    assert np.array_equal(got["n_basis"], G["store_n_basis"][..., :n_iter])
    M1.engine.close()


def test_ring_store_with_a_custom_sink_and_overlap(golden):
    """A sink sees every chunk once, in order, on the drain stream; the sampling stream never waits for the host."""
    import torch

    G = golden("gmrf_chain")
    seen = []
    parts = {}

    def sink(key, it0, it1, block):
        seen.append((key, it0, it1, torch.cuda.current_stream().cuda_stream))
        parts.setdefault(key, []).append(block.clone())

    M0, _ = build(G, "sparse_", True, 3, fuse=True, n_burn=0, n_iter=10, seed=2)
    M0.run_mcmc()
    want = M0.collect()
    M0.engine.close()
    M1, _ = build(G, "sparse_", True, 3, fuse=True, n_burn=0, n_iter=10, seed=2, store_ring=4, sink=sink)
    M1.run_mcmc()
    main = M1.engine._stream.cuda_stream
    assert [s[1:3] for s in seen if s[0] == "b"] == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 10)]
    assert all(s[3] != main for s in seen)
    b = torch.cat(parts["b"]).cpu().numpy()  # (n_iter, C, n)
    assert np.array_equal(np.moveaxis(b, 0, -1), want["b"])
    M1.engine.close()


def test_streaming_gather_two_ranks_share_the_gpu(golden, tmp_path):
    """Two ranks (gloo, both on this GPU), each with its shard of the chains and a ring store drained through GatherSink: the
    root's result is the single-process run of all chains, bit for bit (global-chain-id streams + the ring + the collective)."""
    script = os.path.join(ROOT, "tests", "native", "streaming_gather_rank.py")
    out = str(tmp_path / "gathered.npz")
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OUT=out)
    procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r)), cwd=ROOT, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = np.load(out)
    G = golden("gmrf_chain")
    M0, _ = build(G, "sparse_", True, 5, fuse=True, n_burn=2, n_iter=11, seed=4)
    M0.run_mcmc()
    want = M0.collect()
    M0.engine.close()
    for key in want:
        assert np.array_equal(got[key], want[key]), key
    assert np.array_equal(got["thin_b"], want["b"][..., ::3])


@pytest.mark.parametrize("fuse", [True, False])
def test_ring_store_with_thinning(golden, fuse):
    """n_thin = 3: only every third sweep is stored (mcmc.py:97-106); the ring holds stored iterations, not sweeps."""
    G = golden("gmrf_chain")
    outs = []
    for ring in (0, 4):
        M, _ = build(G, "sparse_", True, 4, fuse=fuse, n_burn=2, n_iter=9, seed=6, store_ring=ring)
        M.n_thin = 3
        M.run_mcmc()
        outs.append(M.collect())
        M.engine.close()
    for key in outs[0]:
        assert np.array_equal(outs[0][key], outs[1][key]), key


def test_ring_store_keeps_the_fitted_values_of_a_regression(golden):
    """Example 3 (linear regression) stores the response's fitted values next to the parameters (mcmc.py:109-111): they ride the ring
    like every other entry."""
    from test_mcmc_api_gpu import build_linreg

    G = golden("linreg_chain")
    outs = []
    for ring in (0, 6):
        M = build_linreg(G, "ex3_", 3, store_ring=ring)
        M.run_mcmc()
        outs.append(M.collect())
        M.engine.close()
    assert "y" in outs[0]
    for key in outs[0]:
        assert np.array_equal(outs[0][key], outs[1][key], equal_nan=True), key


def test_quantiles_of_a_ragged_store_ignore_the_padding(golden):
    """The reversible-jump model's theta / beta stores are NaN beyond the live length: MCMC.quantiles leaves the padding out per
    element like np.nanquantile on the collected store (an element that no iteration of a chain ever had reads NaN)."""
    import warnings

    from test_rj_chain_gpu import run_with_tape

    G = golden("rj_gmrf_chain")
    chains = np.arange(G["init_k"].shape[0])
    M, _, _ = run_with_tape(G, chains, 60)
    M.run_mcmc()
    out = M.collect()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for key in ("theta", "beta", "n_basis", "lambda"):
            arr = out[key].reshape(out[key].shape[0], -1, out[key].shape[-1])          # (C, size, n_iter)
            got = M.quantiles(key, [0.1, 0.5, 0.9], pooled=False)
            assert same(got, np.nanquantile(arr, [0.1, 0.5, 0.9], axis=2)), key
            flat = np.transpose(arr, (1, 0, 2)).reshape(arr.shape[1], -1)
            assert same(M.quantiles(key, [0.1, 0.5, 0.9], pooled=True), np.nanquantile(flat, [0.1, 0.5, 0.9], axis=1)), key
    M.engine.close()
