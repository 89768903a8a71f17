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


def test_default_build_carries_no_ablation_variants(built_lib):
    """VERDICT r3 hygiene: the default build compiles only the kernels the data path can reach.  The tuning keys that select
    variant / ablation instantiations are refused (with the rebuild recipe) unless the library was built with
    -DMANET_ABLATION; the keys that only steer the shipped kernels (block map, split count, the unfused local route) work."""
    import subprocess as sp
    import sys
    code = (
        "import os, sys; os.environ['MANET_TUNING']='1'; sys.path.insert(0, %r)\n"
        "from cvpr2020_manet_amd import _lib\n"
        "lib = _lib.load()\n"
        "assert lib.manet_tune_set(0, 4) == 0 and lib.manet_tune_set(0, -2**31) == 0\n"
        "assert lib.manet_tune_set(1, 16) == 0 and lib.manet_tune_set(1, -2**31) == 0\n"
        "res = [lib.manet_tune_set(k, 1) for k in (2, 3, 5, 6)]\n"
        "msg = lib.manet_last_error_string().decode()\n"
        "print(res, msg)\n" % ROOT)
    out = sp.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    nm = sp.run(["nm", "-C", "--defined-only", built_lib], capture_output=True, text=True).stdout
    ablation_build = "global_match_bf16_wide_kernel<7, 15" in nm
    if ablation_build:
        assert "[0, 0, 0, 0]" in out.stdout
    else:
        assert "[-1, -1, -1, -1]" in out.stdout and "MANET_ABLATION" in out.stdout
        # none of the spilled / unreachable instantiations r3's build carried
        for sym in ("global_match_bf16_wide_kernel<9", "global_match_bf16_wide_kernel<7, 4", "global_match_bf16_kernel<9, true, 2",
                    "global_match_f32_kernel<64, 1, false>", "frame_prepare_kernel<float, 64>"):
            assert sym not in nm, sym
