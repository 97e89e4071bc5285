"""CPU: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/gemnet_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from gemnet_pytorch_amd import _lib


def declared_symbols():
    with open(os.path.join(ROOT, "include", "gemnet_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gn_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return _lib.load()


def test_header_declares_the_bound_functions():
    syms = declared_symbols()
    assert set(_lib.SIGNATURES) <= set(syms)
    assert {"gn_abi_version", "gn_error_string"} <= set(syms)


def test_library_exports_every_declared_symbol(lib):
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/gemnet_hip.h but not exported"
    assert lib.gn_abi_version() == 15


def test_gemm_args_struct_matches_header():
    with open(os.path.join(ROOT, "include", "gemnet_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = re.search(r"typedef struct \{(.*?)\} gn_gemm_args;", text, flags=re.S).group(1)
    names = re.findall(r"[\*\s]([A-Za-z_][A-Za-z0-9_]*)\s*[;,]", body)
    assert names == [n for n, _ in _lib.GemmArgs._fields_]
