"""Compile-time guard for the sampling kernels' private memory (no GPU needed: hipcc cross-compiles).

A by-value kernel argument that is indexed with something the compiler cannot resolve at compile time gets a private copy
in scratch; that happened to the generic tridiagonal instantiation in round 2 (1.4 KB + 360 bytes per lane, 500 instead of
110 us per sweep) without any test noticing.  The workgroup-per-chain instantiations the headline sizes use must not need
scratch at all."""

import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_tridiagonal_kernels_need_no_scratch():
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", f"-I{ROOT}/include", "-c",
           f"{ROOT}/openmcmc_amd/csrc/omc_tridiag.hip", "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    text = out.stderr + out.stdout
    scratch = {}
    name = None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            scratch[name] = int(m.group(1))
    wanted = [k for k in scratch if re.match(r"_Z13k_tridiag_segILi(8|10|16|20)ELb1ELi\d+ELi[01]EEv7TriArgsi", k)]
    assert len(wanted) >= 6, sorted(scratch)  # M = 8, 10 in both forms, 16 and 20 generic
    bad = {k: scratch[k] for k in wanted if scratch[k] != 0}
    assert not bad, bad
