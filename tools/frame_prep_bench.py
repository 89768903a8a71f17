#!/usr/bin/env python3
"""Times manet_frame_prepare against the two passes it replaces (query pack + pooling pass) at the BASELINE grids.
Run on the GPU box: python3 tools/frame_prep_bench.py   (never put this script itself after `rocprofv3 --`)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvpr2020_manet_amd import _lib, ops  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def kernel_us(lib, fn, channel, n=40):
    """HIP events the library records right around the kernel (manet_profile_*): the launch's own duration -- the Python
    call above it costs more than the kernel (host-bound loop), so wall / loop timing says nothing about it"""
    import ctypes
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    _lib.check(lib.manet_profile_begin(n + 1), "manet_profile_begin")
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    ms, cnt = (ctypes.c_float * (n + 1))(), ctypes.c_int(0)
    _lib.check(lib.manet_profile_read(channel, ms, n + 1, ctypes.byref(cnt)), "manet_profile_read")
    vals = sorted(ms[i] for i in range(cnt.value))
    _lib.check(lib.manet_profile_end(None, 0, None), "manet_profile_end")
    return vals[len(vals) // 2] * 1e3 if vals else float("nan")


def main():
    os.environ["MANET_TUNING"] = "1"
    lib = _lib.load()
    for (h, w, d, compute, st) in ((120, 214, 12, "f32", torch.float32), (120, 214, 4, "bf16", torch.bfloat16),
                                   (180, 320, 4, "bf16", torch.bfloat16), (120, 214, 4, "f32", torch.bfloat16)):
        e = (torch.relu(torch.randn(2, 100, h, w, device="cuda")) * 0.1).to(st)
        lab = torch.zeros(h, w, dtype=torch.int32, device="cuda")
        for xc in (0, 1):
            if lib.manet_tune_set(6, xc) != 0:  # (the 64-column variant lives in -DMANET_ABLATION builds only)
                if xc:
                    continue
            t_prep = timeit(lambda: ops.prepare_frames(e[0], compute=compute, max_distance=d))
            t_prep_nopool = timeit(lambda: ops.prepare_frames(e[0], compute=compute, max_distance=-1))
            k_prep = kernel_us(lib, lambda: ops.prepare_frames(e[0], compute=compute, max_distance=d), 2)
            k_nopool = kernel_us(lib, lambda: ops.prepare_frames(e[0], compute=compute, max_distance=-1), 2)
            print("%dx%d d=%d %s/%s XC=%d: frame_prepare KERNEL %.1f us (image only %.1f us); python loop %.1f / %.1f us per call"
                  % (h, w, d, compute, st, 64 if xc else 32, k_prep, k_nopool, t_prep, t_prep_nopool))
        t_pack = timeit(lambda: ops.PackedQuery(e[0].permute(1, 2, 0), compute=compute))
        fr = ops.prepare_frames(e, compute=compute, max_distance=d)
        t_old = timeit(lambda: ops.local_match(e[0].permute(1, 2, 0), e[1].permute(1, 2, 0), lab, 2, d))
        t_new = timeit(lambda: ops.local_match_frames(fr[0], fr[1], lab, 2))
        print("   query pack %.1f us; local match: pooling pass + fused %.1f us, fused alone %.1f us" % (t_pack, t_old, t_new))


if __name__ == "__main__":
    main()
