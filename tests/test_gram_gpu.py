"""G = X' diag(w) X on the fp64 matrix cores (omc_gram -> k_gram_mfma): against NumPy's float64 product and against
the rocBLAS route of the same entry point, on shapes that exercise partial tiles, single and several contraction
slices, and the asymmetric-operand check the MFMA lane maps call for.  Reference call site: location_scale.py:238-241."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.max(np.abs(a - b)) / np.max(np.abs(b))


@pytest.mark.parametrize("n,p,weighted", [(53, 37, True), (64, 16, False), (1000, 129, True), (777, 300, True),
                                          (4096, 256, False), (10000, 1000, True)])
def test_gram_matches_numpy_and_rocblas(n, p, weighted):
    from openmcmc_amd.engine import Engine

    eng = Engine(1)
    rng = np.random.default_rng(n + p)
    X = rng.standard_normal((n, p)) * (1 + np.arange(p) / p)   # columns of different scale: a transposed tile would show
    w = 0.5 + rng.random(n) if weighted else None
    dX, dw = eng.to_device(X), (eng.to_device(w) if weighted else None)
    G = eng.gram(dX, dw).cpu().numpy()
    exp = (X.T * w) @ X if weighted else X.T @ X
    assert rel(G, exp) < 1e-13
    assert np.array_equal(G, G.T)                               # mirrored, not recomputed
    eng.set_option("gram_use_rocblas", 1)
    Gb = eng.gram(dX, dw).cpu().numpy()
    assert rel(G, Gb) < 1e-13
    eng.set_option("gram_use_rocblas", 0)
    G2 = eng.gram(dX, dw).cpu().numpy()
    assert np.array_equal(G, G2)                                # fixed summation order: bit-reproducible
    eng.close()


def test_gram_exact_on_integer_data():
    """Small integers: every product and sum is exact in fp64, so the lane maps of v_mfma_f64_16x16x4_f64 are checked bit
    for bit (an off-by-one row map would still pass a tolerance test on random data)."""
    from openmcmc_amd.engine import Engine

    eng = Engine(1)
    rng = np.random.default_rng(1)
    n, p = 200, 150
    X = rng.integers(-8, 9, size=(n, p)).astype(float)
    w = rng.integers(1, 5, size=n).astype(float)
    G = eng.gram(eng.to_device(X), eng.to_device(w)).cpu().numpy()
    assert np.array_equal(G, (X.T * w) @ X)
    eng.close()
