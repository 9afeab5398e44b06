"""Compile-time guards for the sampling kernels (no GPU needed: hipcc cross-compiles).

1. Private memory.  A by-value kernel argument that is indexed with something the compiler cannot resolve at compile time
   gets a private copy in scratch; that happened to the generic tridiagonal instantiation in round 2 (1.4 KB + 360 bytes
   per lane, 500 instead of 110 us per sweep) without any test noticing.  The workgroup-per-chain instantiations the
   headline sizes use must not need scratch at all.

2. Entry state of the self-restarting workgroups.  omc_gmrf_run's default launch form ends a sweep with `s_setpc_b64` to the
   kernel's first instruction after setting s[0:1] (kernel-argument pointer), s2 (workgroup id) and v0 (work-item id) by
   hand (omc_tridiag.hip, OMC_REENTER).  That reproduces a fresh workgroup only while the kernel descriptor asks the
   dispatcher for exactly those registers.  The descriptors of the two re-entered instantiations are taken from the code
   object this source compiles to and handed to the library's own test (omc_reentry_descriptor_ok -- the one the library
   applies at run time to the descriptors it reads back from the device); deliberately altered descriptors must fail it."""

import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
LLVM = "/opt/rocm/lib/llvm/bin"
REENTERED = ["_Z13k_tridiag_segILi8ELb1ELi1024ELi1EEv7TriArgsi", "_Z13k_tridiag_segILi10ELb1ELi1024ELi1EEv7TriArgsi"]

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")


@pytest.fixture(scope="module")
def compiled(tmp_path_factory):
    """omc_tridiag.hip compiled once: (resource-usage remarks, path of the unbundled gfx950 code object)"""
    d = tmp_path_factory.mktemp("tridiag")
    obj = str(d / "omc_tridiag.o")
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", f"-I{ROOT}/include",
           "-mllvm", "-instcombine-max-copied-from-constant-users=100000",  # as openmcmc_amd/csrc/Makefile (see the note there)
           "-c",
           f"{ROOT}/openmcmc_amd/csrc/omc_tridiag.hip", "-o", obj, "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    fat, co = str(d / "fat.bin"), str(d / "tridiag.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], check=True, timeout=120)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={co}"], check=True, timeout=120)
    return out.stderr + out.stdout, co


def kernel_descriptor(co, kernel):
    """The 64 bytes of `kernel`.kd from an AMDGPU code object (ELF64 little endian), via the symbol and section tables."""
    data = open(co, "rb").read()
    assert data[:4] == b"\x7fELF" and data[4] == 2 and data[5] == 1
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    symtab = next(s for s in secs if s[1] == 2)  # SHT_SYMTAB
    strtab = secs[symtab[6]]
    want = (kernel + ".kd").encode()
    for off in range(symtab[4], symtab[4] + symtab[5], 24):
        name, info, other, shndx, value, size = struct.unpack_from("<IBBHQQ", data, off)
        end = data.index(b"\0", strtab[4] + name)
        if data[strtab[4] + name:end] == want:
            sec = secs[shndx]
            assert size == 64
            return data[sec[4] + (value - sec[3]): sec[4] + (value - sec[3]) + 64]
    raise AssertionError(f"{kernel}.kd not in the code object")


def descriptor_words(kd):
    private_size, = struct.unpack_from("<I", kd, 4)
    rsrc2, props_preload = struct.unpack_from("<II", kd, 52)
    return private_size, rsrc2, props_preload


def test_tridiagonal_kernels_need_no_scratch(compiled):
    text, _ = compiled
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
    # the waiting form of the smoother kernel (SIG 2) never restarts itself, so a few spilled registers are legal there --
    # a private copy of an argument block (hundreds of bytes) is not
    waiting = {k: v for k, v in scratch.items() if re.match(r"_Z13k_tridiag_segILi(8|10)ELb1ELi\d+ELi2EEv7TriArgsi", k)}
    assert len(waiting) == 2 and all(v <= 64 for v in waiting.values()), waiting


def test_reentered_kernels_ask_for_the_entry_state_the_restart_sets(compiled):
    from openmcmc_amd import _abi

    _, co = compiled
    for kernel in REENTERED:
        kd = kernel_descriptor(co, kernel)
        private_size, rsrc2, pp = descriptor_words(kd)
        # spelled out once here, independently of the library's test
        assert private_size == 0 and (rsrc2 & 1) == 0, "private segment"
        assert (rsrc2 >> 1) & 0x1F == 2, "user SGPR count"
        assert pp & 0x7F == 0x08, "user SGPRs other than the kernel-argument pointer"
        assert (rsrc2 >> 7) & 0xF == 0x1, "system SGPRs other than workgroup id x"
        assert pp >> 16 == 0, "kernel-argument preload"
        assert (pp >> 10) & 1 == 0 and (pp >> 11) & 1 == 0, "wave32 / dynamic stack"
        assert _abi.lib.omc_reentry_descriptor_ok(private_size, rsrc2, pp) == 1


def test_the_descriptor_check_rejects_other_entry_states(compiled):
    from openmcmc_amd import _abi

    _, co = compiled
    private_size, rsrc2, pp = descriptor_words(kernel_descriptor(co, REENTERED[1]))
    ok = _abi.lib.omc_reentry_descriptor_ok
    assert ok(private_size, rsrc2, pp) == 1
    altered = {
        "four user SGPRs (dispatch pointer on)": (private_size, (rsrc2 & ~(0x1F << 1)) | (4 << 1), pp | 0x02),
        "queue pointer": (private_size, (rsrc2 & ~(0x1F << 1)) | (4 << 1), pp | 0x04),
        "dispatch id": (private_size, (rsrc2 & ~(0x1F << 1)) | (4 << 1), pp | 0x10),
        "flat scratch init": (private_size, (rsrc2 & ~(0x1F << 1)) | (4 << 1), pp | 0x20),
        "two preloaded kernel arguments": (private_size, (rsrc2 & ~(0x1F << 1)) | (4 << 1), pp | (2 << 16)),
        "preload length alone": (private_size, rsrc2, pp | (1 << 16)),
        "workgroup id y": (private_size, rsrc2 | (1 << 8), pp),
        "workgroup id z": (private_size, rsrc2 | (1 << 9), pp),
        "workgroup info": (private_size, rsrc2 | (1 << 10), pp),
        "no workgroup id x": (private_size, rsrc2 & ~(1 << 7), pp),
        "private segment enabled": (private_size, rsrc2 | 1, pp),
        "scratch bytes": (64, rsrc2, pp),
        "dynamic stack": (private_size, rsrc2, pp | (1 << 11)),
        "wave32": (private_size, rsrc2, pp | (1 << 10)),
        "user SGPR count only": (private_size, (rsrc2 & ~(0x1F << 1)) | (3 << 1), pp),
    }
    for what, words in altered.items():
        assert ok(*[w & 0xFFFFFFFF for w in words]) == 0, what


def test_generic_kernel_of_the_same_file_differs_where_expected(compiled):
    """The parser reads the right bytes: the serial kernel is a different function with its own descriptor (a kernel that
    takes no LDS), and the LDS size of the headline instantiation is the one DESIGN.md states (the whole CU)."""
    _, co = compiled
    kd = kernel_descriptor(co, REENTERED[1])
    lds, = struct.unpack_from("<I", kd, 0)
    assert 150_000 < lds <= 163_840
    kd_serial = kernel_descriptor(co, "_Z16k_tridiag_serial7TriArgs")
    assert struct.unpack_from("<I", kd_serial, 0)[0] == 0
