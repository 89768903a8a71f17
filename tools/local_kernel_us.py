#!/usr/bin/env python3
"""The fused local-window kernel's own duration (HIP events the library records around it, manet_profile channel 1) at
the BASELINE grids: python3 tools/local_kernel_us.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MANET_TUNING"] = "1"
import torch  # noqa: E402

from cvpr2020_manet_amd import _lib, ops  # noqa: E402


def kernel_us(lib, fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    _lib.check(lib.manet_profile_begin(n + 1), "begin")
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    ms, cnt = (ctypes.c_float * (n + 1))(), ctypes.c_int(0)
    _lib.check(lib.manet_profile_read(1, ms, n + 1, ctypes.byref(cnt)), "read")
    vals = sorted(ms[i] for i in range(cnt.value))
    _lib.check(lib.manet_profile_end(None, 0, None), "end")
    return vals[len(vals) // 2] * 1e3


def main():
    lib = _lib.load()
    for (h, w, d, nid) in ((120, 214, 4, 4), (120, 214, 12, 2), (180, 320, 4, 6), (120, 214, 2, 2), (120, 214, 8, 2)):
        e = torch.relu(torch.randn(2, 100, h, w, device="cuda")) * 0.1
        lab = torch.randint(0, nid, (h, w), device="cuda", dtype=torch.int32)
        fr = ops.prepare_frames(e, compute="f32", max_distance=d)
        out = torch.empty(h, w, nid, device="cuda")
        t = kernel_us(lib, lambda: ops.local_match_frames(fr[0], fr[1], lab, nid, out=out))
        print("%dx%d d=%d n_ids=%d: fused kernel %.1f us" % (h, w, d, nid, t))


if __name__ == "__main__":
    main()
