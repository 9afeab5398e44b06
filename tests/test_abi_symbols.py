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


def test_terms_struct_layout_matches_header():
    from openmcmc_amd import _abi

    # int32 + padding, then 5 arrays of 4 pointers
    assert ctypes.sizeof(_abi.TridiagTerms) == 8 + 5 * 4 * 8
    assert _abi.TridiagTerms.diag.offset == 8


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "openmcmc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
