"""CPU: the C-ABI library loads and exports every symbol that include/manet_hip.h declares
(no compute calls -- there is no GPU here)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def built_lib():
    so = os.path.join(ROOT, "cvpr2020_manet_amd", "libmanet_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "cvpr2020_manet_amd", "csrc")])
    return so


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "manet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(manet_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for s in ["manet_global_match", "manet_bank_prepare", "manet_global_match_prepared",
              "manet_local_match_f32", "manet_local_dist_f32", "manet_correlation_forward_f32",
              "manet_normalize_merge_f32", "manet_last_error_string"]:
        assert s in syms


def test_library_exports_every_declared_symbol(built_lib):
    from cvpr2020_manet_amd import _lib
    lib = _lib.load()  # binds every entry of SIGNATURES; raises if one is missing
    syms = declared_symbols()
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header are out of sync"
    for s in syms:
        assert hasattr(lib, s)
    assert b"gfx950" in lib.manet_version()


def test_argument_validation_without_a_gpu(built_lib):
    """pure host-side checks of the ABI: size queries and error reporting need no device"""
    import ctypes
    from cvpr2020_manet_amd import _lib
    lib = _lib.load()
    n = ctypes.c_size_t(0)
    assert lib.manet_global_match_workspace_bytes(25680, 128400, 100, 2, 1, _lib.COMPUTE_F32, ctypes.byref(n)) == 0
    assert 50e6 < n.value < 200e6  # packed bank (128400 x 104 floats) + packed queries + keys
    assert lib.manet_global_match_workspace_bytes(10, 10, 1000, 2, 1, 0, ctypes.byref(n)) == -1
    assert b"C=1000" in lib.manet_last_error_string()
    assert lib.manet_local_workspace_bytes(120, 214, 100, 13, 1, ctypes.byref(n)) == -1
    oc, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.manet_correlation_out_dims(20, 30, 4, 1, 4, 1, 1, ctypes.byref(oc), ctypes.byref(oh), ctypes.byref(ow)) == 0
    assert (oc.value, oh.value, ow.value) == (81, 20, 30)
