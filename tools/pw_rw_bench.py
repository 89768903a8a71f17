#!/usr/bin/env python3
"""The exact-fp32 1x1 layer: resident-weights kernel (r4) against the LDS-weights kernel (r3) and the framework's GEMM.
HIP-event time of back-to-back launches (no Python between them matters at ~100 us per launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MANET_TUNING"] = "1"
import torch  # noqa: E402

from cvpr2020_manet_amd import _lib, ops  # noqa: E402


def timeit(fn, n=30):
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


def main():
    lib = _lib.load()
    for (B, cin, h, w) in ((3, 256, 120, 214), (2, 256, 120, 214), (6, 256, 180, 320), (1, 256, 120, 214)):
        x = torch.randn(B, cin, h, w, device="cuda")
        w2t = torch.randn(cin, 256, device="cuda") * 0.1
        b2 = torch.randn(256, device="cuda")
        wconv = w2t.t().reshape(256, cin, 1, 1).contiguous()
        flop = 2.0 * B * h * w * cin * 256
        lib.manet_tune_set(8, 1)
        t_old = timeit(lambda: ops.conv1x1_mfma(x, w2t, b2))
        lib.manet_tune_set(8, -2 ** 31)
        t_new = timeit(lambda: ops.conv1x1_mfma(x, w2t, b2))
        t_fw = timeit(lambda: torch.nn.functional.conv2d(x, wconv, b2))
        sw = ops.SplitWeight(w2t)
        t_x3 = timeit(lambda: ops.conv1x1_split(x, sw, b2))
        print("[%d,%d,%d,%d]: resident weights %.1f us = %.1f TFLOP/s (%.2f of 157.3); LDS weights %.1f us (%.2f); framework GEMM %.1f us; "
              "split-bf16 %.1f us" % (B, cin, h, w, t_new, flop / t_new / 1e6, flop / t_new / 1e6 / 157.3, t_old, flop / t_old / 1e6 / 157.3,
                                      t_fw, t_x3))


if __name__ == "__main__":
    main()
