"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol
that include/omcmc_hip.h declares, with the binding's signatures in step with the header."""

import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "omcmc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(omc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from openmcmc_amd import _abi

    names = header_functions()
    assert len(names) >= 15
    lib = ctypes.CDLL(_abi.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(_abi.SIGNATURES) == names, "ctypes SIGNATURES out of step with the header"


def test_abi_version_and_error_text_callable_without_gpu():
    from openmcmc_amd import _abi

    assert _abi.lib.omc_abi_version() == 2
    assert isinstance(_abi.lib.omc_last_error(), bytes)


def test_terms_struct_layout_matches_header(tmp_path):
    """ctypes' omc_tridiag_terms against the C compiler's view of include/omcmc_hip.h (ABI 2: center_chain, ld_center_chain)."""
    import shutil
    import subprocess

    from openmcmc_amd import _abi

    T = _abi.TridiagTerms
    # int32 + padding, 5 arrays of 4 pointers, the per-chain centres (4 pointers) and their row stride
    assert ctypes.sizeof(T) == 8 + 5 * 4 * 8 + 4 * 8 + 8
    assert T.diag.offset == 8 and T.center_chain.offset == 8 + 5 * 32 and T.ld_center_chain.offset == 8 + 6 * 32
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        return
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "omcmc_hip.h"\n'
        'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(omc_tridiag_terms), offsetof(omc_tridiag_terms, diag),'
        ' offsetof(omc_tridiag_terms, off), offsetof(omc_tridiag_terms, rhs), offsetof(omc_tridiag_terms, center),'
        ' offsetof(omc_tridiag_terms, scale), offsetof(omc_tridiag_terms, center_chain), offsetof(omc_tridiag_terms, ld_center_chain));'
        ' return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.run([cc, "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True, timeout=120)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True, timeout=30).stdout.split()]
    want = [ctypes.sizeof(T), T.diag.offset, T.off.offset, T.rhs.offset, T.center.offset, T.scale.offset, T.center_chain.offset,
            T.ld_center_chain.offset]
    assert got == want, (got, want)


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "openmcmc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
