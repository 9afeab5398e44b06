"""Random streams without a GPU: the Philox model against the Random123 known-answer vectors, and the host build
of the device's words -> N(0,1) map (omc_common.h: omc_normal_pair) against the NumPy model."""

import os
import shutil
import subprocess

import numpy as np
import pytest
from scipy import stats

import philox_model as pm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Random123 (Salmon et al., SC'11) kat_vectors, philox4x32 10 rounds: counter, key -> output
KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


@pytest.mark.parametrize("ctr,key,expect", KAT)
def test_philox4x32_10_known_answers(ctr, key, expect):
    out = pm.philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], key[0], key[1])
    assert tuple(int(v[0]) for v in out) == expect


@pytest.fixture(scope="module")
def host_exe(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path_factory.mktemp("native") / "normal_pair_host")
    subprocess.run([hipcc, "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "openmcmc_amd", "csrc"), os.path.join(ROOT, "tests", "native", "normal_pair_host.hip"),
                    "-o", exe], check=True)
    return exe


def run_host(exe, words):
    out = subprocess.run([exe], input=np.ascontiguousarray(words, dtype=np.uint32).tobytes(), capture_output=True, check=True).stdout
    return np.frombuffer(out, dtype=np.float64).reshape(-1, 2)


def test_normal_pair_map_matches_model(host_exe):
    rng = np.random.default_rng(0)
    w = rng.integers(0, 2**32, size=(200000, 4), dtype=np.uint32)
    w[0] = 0                                      # u = 1: radius 0
    w[1] = 0xFFFFFFFF                             # smallest u (2^-52), largest angle
    w[2] = (0, 0, 0xFFFFFFFF, 0xFFFFFFFF)
    w[3] = (0xFFFFFFFF, 0xFFFFFFFF, 0, 0)
    w[4:1028, 3] = (np.arange(1024, dtype=np.uint32) << np.uint32(2)) | (w[4:1028, 3] & np.uint32(0xFFFFF003))  # every sign/swap/low-bit pattern
    z = run_host(host_exe, w)
    e0, e1 = pm.normal_pairs(w[:, 0], w[:, 1], w[:, 2], w[:, 3])
    assert np.all(np.isfinite(z))
    assert np.max(np.abs(z[:, 0] - e0)) < 1e-14 and np.max(np.abs(z[:, 1] - e1)) < 1e-14
    assert abs(z[1, 0]) == pytest.approx(np.sqrt(2 * 52 * np.log(2)) * np.sin(np.pi / 4), rel=1e-9)


def test_normal_pair_map_is_standard_normal(host_exe):
    """Moments, Kolmogorov-Smirnov, independence of the two outputs, tail mass, and uniformity of the angle."""
    x, y, z, w = pm.rng_blocks(99, 3, "normal", 17, np.arange(500000))
    n = run_host(host_exe, np.stack([x, y, z, w], axis=1))
    f = n.reshape(-1)
    assert abs(f.mean()) < 4 / np.sqrt(f.size) and abs(f.var() - 1) < 5e-3
    assert stats.kstest(f, "norm").pvalue > 1e-3
    assert abs(np.corrcoef(n[:, 0], n[:, 1])[0, 1]) < 5e-3
    assert abs(np.corrcoef(n[:, 0] ** 2, n[:, 1] ** 2)[0, 1]) < 5e-3
    tail = np.mean(np.abs(f) > 3.0)
    assert abs(tail - 2 * stats.norm.sf(3.0)) < 5 * np.sqrt(2 * stats.norm.sf(3.0) / f.size)
    ang = np.arctan2(n[:, 1], n[:, 0])
    assert stats.kstest(ang, "uniform", args=(-np.pi, 2 * np.pi)).pvalue > 1e-3
    r2 = (n ** 2).sum(1)
    assert stats.kstest(r2, "chi2", args=(2,)).pvalue > 1e-3


def test_inverse_normal_cdf_as241_host_build(tmp_path):
    """omc_ndtri_as241 (omc_truncnorm.h), the fast path of the truncated-normal inverse CDF, built for the host: against
    scipy.special.ndtri over the whole open interval, tails down to 1e-300 and up to 1 - 1e-15."""
    from scipy.special import ndtri

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "ndtri_host")
    subprocess.run([hipcc, "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "openmcmc_amd", "csrc"), os.path.join(ROOT, "tests", "native", "ndtri_host.hip"),
                    "-o", exe], check=True)
    rng = np.random.default_rng(0)
    p = np.concatenate([rng.random(100000), 10.0 ** (-rng.uniform(0, 300, 10000)), 1 - 10.0 ** (-rng.uniform(0, 15, 10000)),
                        [0.5, 0.075, 0.925, 0.0749999, 1e-15, 1 - 1e-15]])
    p = p[(p > 0) & (p < 1)]
    out = np.frombuffer(subprocess.run([exe], input=p.tobytes(), capture_output=True, check=True).stdout, dtype=np.float64)
    ref = ndtri(p)
    err = np.abs(out - ref) / np.maximum(np.abs(ref), 1e-3)
    assert err.max() < 4e-15
