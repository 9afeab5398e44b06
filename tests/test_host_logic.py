"""CPU tests of the host side above the C ABI: state coercion, model algebra, structure
recognition, the RW1 builders (bit-exact against the reference's output) and sharding."""

import numpy as np
import pytest
from scipy import sparse

from openmcmc_amd import gmrf
from openmcmc_amd.chains import host_2d
from openmcmc_amd.distribution.distribution import Gamma
from openmcmc_amd.distribution.location_scale import Normal, tridiagonal_bands
from openmcmc_amd.model import Model
from openmcmc_amd.parallel import shard_chains, store_to_reference_layout
from openmcmc_amd.parameter import Identity, LinearCombination, ScaledMatrix, _is_identity


@pytest.mark.parametrize("n", [1, 2, 5, 40])
def test_precision_irregular_matches_reference(golden, n):
    G = golden("precision_builders")
    P = gmrf.precision_irregular(G[f"irr{n}_s"])
    P = P.toarray() if sparse.issparse(P) else np.asarray(P, dtype=float)
    assert np.array_equal(P, G[f"irr{n}_P"])
    assert np.array_equal(np.asarray(gmrf.precision_irregular(G[f"irr{n}_s"], is_sparse=False), dtype=float), G[f"irr{n}_P"])
    if n > 1:  # reference test_precision: symmetric, rows sum to zero
        assert np.allclose(P, P.T) and np.allclose(P.sum(1), 0)


def test_precision_temporal_matches_reference(golden):
    import pandas as pd

    G = golden("precision_builders")
    t = pd.date_range(start="2022-04-01T01:00:00", end="2022-04-01T01:01:00", periods=30)
    assert np.array_equal(gmrf.precision_temporal(t).toarray(), G["temporal_P"])
    assert np.array_equal(gmrf.precision_temporal(t, unit_length=30.0).toarray(), G["temporal_P_unit30"])


def test_state_coercion_like_reference():
    """mcmc.py:69-76: scalars -> (1,1), lists and 1-D arrays -> columns, 2-D untouched."""
    assert host_2d(3).shape == (1, 1) and host_2d(3).dtype == np.float64
    assert host_2d([1, 2, 3]).shape == (3, 1)
    assert host_2d(np.arange(4.0)).shape == (4, 1)
    a = np.ones((2, 3))
    assert host_2d(a) is a


def test_tridiagonal_bands():
    n = 6
    P = sparse.diags((-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)), offsets=[-1, 0, 1], format="csc")
    d, o = tridiagonal_bands(P, n)
    assert np.array_equal(d, 2 * np.ones(n)) and np.array_equal(o, -np.ones(n - 1))
    assert tridiagonal_bands(sparse.identity(n, format="csc"), n) == (None, None)
    d, o = tridiagonal_bands(np.diag(np.arange(1.0, n + 1)), n)
    assert o is None and np.array_equal(d, np.arange(1.0, n + 1))
    wide = P.toarray()
    wide[0, 3] = wide[3, 0] = 0.5
    assert tridiagonal_bands(wide, n) is None
    asym = P.toarray()
    asym[0, 1] = 7.0
    with pytest.raises(ValueError):
        tridiagonal_bands(asym, n)
    assert _is_identity(sparse.identity(n), n) and not _is_identity(P, n)


def example4_model():
    return Model([
        Normal("y", mean="b", precision=ScaledMatrix(matrix="P_tau", scalar="tau")),
        Normal("b", mean="mu", precision=ScaledMatrix(matrix="P_lambda", scalar="lambda")),
        Gamma("lambda", shape="a_lam", rate="b_lam"),
        Gamma("tau", shape="a_tau", rate="b_tau"),
    ])


def test_model_conditional_membership():
    """model.py:41-55 and tests/test_distribution.py:253-270 of the reference."""
    mdl = example4_model()
    assert list(mdl.keys()) == ["y", "b", "lambda", "tau"]
    assert set(mdl.conditional("b").keys()) == {"y", "b"}
    assert set(mdl.conditional("lambda").keys()) == {"b", "lambda"}
    assert set(mdl.conditional("tau").keys()) == {"y", "tau"}
    assert mdl["y"].param_list == ["y", "b", "tau", "P_tau"]
    assert isinstance(mdl["y"].mean, Identity) and mdl["y"].mean.form == "b"
    assert mdl.response is None


def test_parameter_lists_and_type_errors():
    lc = LinearCombination(form={"beta": "X", "alpha": "A"})
    assert lc.get_param_list() == ["beta", "alpha", "X", "A"] and lc.get_grad_param_list() == ["beta", "alpha"]
    sm = ScaledMatrix(matrix="P", scalar="lam")
    assert sm.get_param_list() == ["lam", "P"] and sm.get_grad_param_list() == ["lam"]
    state = {"X": np.arange(6.0).reshape(3, 2), "beta": np.ones((2, 1)), "A": np.eye(3), "alpha": np.ones((3, 1)),
             "P": 2 * np.eye(3), "lam": np.array([[3.0]])}
    assert np.array_equal(lc.predictor(state), state["X"] @ state["beta"] + state["alpha"])
    assert np.array_equal(lc.predictor_conditional(state, term_to_exclude="alpha"), state["X"] @ state["beta"])
    assert np.array_equal(sm.predictor(state), 6 * np.eye(3))
    with pytest.raises(TypeError):
        Normal("y", mean=3.0, precision="P")
    with pytest.raises(TypeError):
        Normal("y", mean="m", precision=LinearCombination(form={"a": "b"}))
    with pytest.raises(TypeError):
        Gamma("g", shape=1.0, rate="r")


def test_normal_structure_recognition():
    n = 5
    mdl = example4_model()
    P = sparse.diags((-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)), offsets=[-1, 0, 1], format="csc")
    state = {"P_lambda": P, "P_tau": sparse.identity(n, format="csc"), "lambda": np.array([[2.0]]), "tau": np.array([[1.0]])}
    st = mdl["b"].structure(state)
    assert st.n == n and st.scale_key == "lambda" and st.n_pos == n and st.off is not None
    st = mdl["y"].structure(state)
    assert st.diag is None and st.off is None and st.scale_key == "tau"
    state["P_lambda"] = np.ones((n, n)) + 5 * np.eye(n)
    assert mdl["b"].structure(state).diag is False  # dense: not served by the tridiagonal path
    state["P_lambda"] = np.ones((n, n + 1))
    with pytest.raises(ValueError):
        mdl["b"].structure(state)


def test_gamma_host_log_p_matches_scipy():
    from scipy import stats

    g = Gamma("lam", shape="a", rate="b")
    state = {"lam": np.array([[2.5]]), "a": np.array([[3.0]]), "b": np.array([[0.7]])}
    assert abs(g.log_p(state) - stats.gamma.logpdf(2.5, 3.0, scale=1 / 0.7)) < 1e-13


def test_shard_chains_partitions_exactly():
    for total, world in [(1024, 8), (1024, 1), (10, 4), (3, 8)]:
        blocks = [shard_chains(total, world, r) for r in range(world)]
        assert sum(b[0] for b in blocks) == total
        off = 0
        for n_local, offset in blocks:
            assert offset == off
            off += n_local
    with pytest.raises(ValueError):
        shard_chains(8, 2, 2)


def test_store_layout():
    a = np.arange(2 * 3 * 4.0).reshape(2, 3, 4)  # (n_iter, C, size)
    out = store_to_reference_layout("b", a)
    assert out.shape == (3, 4, 2) and out[1, 2, 0] == a[0, 1, 2]
    lp = np.arange(6.0).reshape(2, 3)
    assert store_to_reference_layout("log_post", lp).shape == (3, 2, 1)


def test_ragged_chain_arrays_on_cpu_tensors():
    """The ragged description (count key + axis) and the padded constructors; torch CPU tensors stand in for
    device memory (no kernel is called)."""
    import torch

    from openmcmc_amd.chains import ChainArray, ragged_from_lists

    theta = ragged_from_lists([[1.0, 2.0], [3.0], [4.0, 5.0, 6.0]], 4, 1, "n_basis", torch.device("cpu"))
    assert theta.shape == (1, 4) and theta.ragged == ("n_basis", 1)
    assert theta.data[1, 0].tolist() == [3.0, 0.0, 0.0, 0.0]
    beta = ragged_from_lists([[1.0], [2.0, 3.0]], 3, 0, "n_basis", torch.device("cpu"))
    assert beta.shape == (3, 1) and beta.vector().shape == (2, 3)
    state = {"n_basis": ChainArray(torch.tensor([2.0, 1.0, 3.0]).reshape(3, 1, 1))}
    assert theta.count(state).tolist() == [2.0, 1.0, 3.0]
    assert theta.like(theta.data + 1).ragged == theta.ragged
    with pytest.raises(ValueError):
        ragged_from_lists([[1.0, 2.0, 3.0]], 2, 0, "k", torch.device("cpu"))
    with pytest.raises(ValueError):
        ChainArray(torch.zeros(2, 3, 1), ragged=("k", 2))
    # a basis kept column-major per chain: logical (C, n, k) view over (C, k, n) storage
    phys = torch.arange(2 * 3 * 5, dtype=torch.float64).reshape(2, 3, 5)
    B = ChainArray(phys.transpose(1, 2), ragged=("n_basis", 1))
    assert B.shape == (5, 3) and B.storage().data_ptr() == phys.data_ptr() and B.columns().is_contiguous()


def test_uniform_and_poisson_host_values_match_scipy():
    from scipy import stats

    from openmcmc_amd.distribution.distribution import Poisson, Uniform

    u = Uniform("theta", domain_response_lower=np.array([[-10.0]]), domain_response_upper=np.array([[10.0]]))
    state = {"theta": np.array([[1.0, 2.0, -3.0]])}
    assert u.log_p(state) == pytest.approx(3 * -np.log(20.0))                 # distribution.py:436-442
    assert np.allclose(u.log_p(state, by_observation=True), -np.log(20.0) * np.ones(3))
    assert u.log_p_per_replicate(state) == pytest.approx(-np.log(20.0))
    u2 = Uniform("x", domain_response_lower=np.array([0.0, -1.0]), domain_response_upper=np.array([2.0, 3.0]))
    assert u2.domain_range({"x": np.zeros((2, 1))}).ravel().tolist() == [2.0, 4.0]
    assert u2.param_list == ["x"]
    p = Poisson("n", rate="rho")
    st = {"n": np.array([[4.0]]), "rho": np.array([[5.0]])}
    assert p.log_p(st) == pytest.approx(stats.poisson.logpmf(4, 5.0))
    assert p.param_list == ["n", "rho"]
    with pytest.raises(TypeError):
        Poisson("n", rate=3.0)


def test_mixture_parameters_host_predictors():
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.parameter import MixtureParameterMatrix, MixtureParameterVector

    st = {"mu": np.array([[1.0], [5.0]]), "tau": np.array([[0.25], [4.0]]), "alloc": np.array([[0], [1], [1]])}
    v = MixtureParameterVector("mu", "alloc")
    assert v.predictor(st).ravel().tolist() == [1.0, 5.0, 5.0]                 # parameter.py:447
    m = MixtureParameterMatrix("tau", "alloc")
    assert np.array_equal(m.predictor(st).toarray(), np.diag([0.25, 4.0, 4.0]))  # parameter.py:501
    assert v.get_param_list() == ["mu", "alloc"] and v.get_grad_param_list() == ["mu"] and m.get_grad_param_list() == []
    d = Normal("beta", mean=v, precision=m)
    assert d.is_mixture and d.param_list == ["beta", "mu", "alloc", "tau", "alloc"]


def test_samplers_keep_full_model_like_reference():
    """RandomWalk with a state_update_function and ReversibleJump keep the WHOLE model
    (metropolis_hastings.py:201-210, reversible_jump.py:66-74); without the callback RandomWalk conditions."""
    import sys, os

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from openmcmc_amd.distribution.distribution import Gamma, Poisson, Uniform
    from openmcmc_amd.distribution.location_scale import Normal
    from openmcmc_amd.model import Model
    from openmcmc_amd.sampler.metropolis_hastings import RandomWalk, RandomWalkLoop
    from openmcmc_amd.sampler.reversible_jump import ReversibleJump

    mdl = Model([Normal("x", mean="mu", precision="Q"), Uniform("theta"), Poisson("n", rate="rho"), Gamma("tau", shape="a", rate="b")])
    assert set(RandomWalk("theta", mdl).model.keys()) == {"theta"}
    assert set(RandomWalkLoop("theta", mdl, state_update_function=lambda s, j: (s, 0.0, 0.0)).model.keys()) == set(mdl.keys())
    rj = ReversibleJump("n", mdl, associated_params="theta", n_max=5)
    assert set(rj.model.keys()) == set(mdl.keys()) and rj.associated_params == ["theta"]
    assert rj.accept_rate.get_acceptance_rate() == "No proposals"
    with pytest.raises(RuntimeError):
        rj.sample({})  # not bound to an engine: the product path fails loudly


def test_null_distribution_and_band_recognition():
    """NullDistribution's host behaviour (location_scale.py:63-124) and the band-structure recognition that routes a
    precision wider than tridiagonal to the band kernel (or to the dense route when the band would not pay)."""
    from scipy import sparse

    from openmcmc_amd.distribution.location_scale import Normal, NullDistribution, band_storage
    from openmcmc_amd.parameter import LinearCombination, ScaledMatrix

    nd = NullDistribution("y", mean=LinearCombination({"beta": "B"}), precision=ScaledMatrix("P", "tau"))
    assert nd.log_p({}) == 0.0 and nd.rvs({}) is None and nd.grad_log_p_diag({}, "beta", None) is None
    assert nd.param_list == ["y", "beta", "B", "tau", "P"]
    n = 12
    D = sparse.diags([np.ones(n - 2), -2 * np.ones(n - 2), np.ones(n - 2)], offsets=[0, 1, 2], shape=(n - 2, n))
    P = (D.T @ D + 1e-3 * sparse.identity(n)).tocsc()
    band = band_storage(P, n)
    assert band.shape == (3, n) and np.array_equal(band[0], P.diagonal()) and np.array_equal(band[2, : n - 2], P.diagonal(-2))
    assert band[2, n - 2:].tolist() == [0.0, 0.0]
    st = Normal("b", mean="mu", precision=ScaledMatrix("P", "lam")).structure({"P": P, "lam": 1.0})
    assert st.diag is False and st.band is not None and st.band_rows().shape == (3, n)
    assert band_storage(np.ones((6, 6)) + 6 * np.eye(6), 6) is None          # dense: the band would not pay
    tri = Normal("b", mean="mu", precision="Q").structure({"Q": sparse.diags([[-1.0] * 4, [2.0] * 5, [-1.0] * 4], [-1, 0, 1]).tocsc()})
    assert tri.band is None and tri.band_rows().shape == (2, 5)
    with pytest.raises(ValueError):
        band_storage(sparse.csc_matrix(np.triu(np.ones((8, 8)), 0) * (np.abs(np.subtract.outer(np.arange(8), np.arange(8))) <= 2)), 8)
