#!/usr/bin/env python3
"""The exact-fp32 1x1 layer: resident-weights kernel against the LDS-weights kernel, the framework's GEMM and the split-bf16 kernel.
Per shape: a long warm-up (the first launches of a process run at ramping clocks: r4's 5 + 30 launches read 104-107 us where a
warm GPU gives ~95), then the forms alternate four times, 60 launches each; the minimum of a form's means is printed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MANET_TUNING"] = "1"
import torch  # noqa: E402

from cvpr2020_manet_amd import _lib, ops  # noqa: E402


def mean_us(fn, n=60):
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
        sw = ops.SplitWeight(w2t)
        flop = 2.0 * B * h * w * cin * 256

        def rw():
            lib.manet_tune_set(8, -2 ** 31)
            return ops.conv1x1_mfma(x, w2t, b2)

        def lds():
            lib.manet_tune_set(8, 1)
            return ops.conv1x1_mfma(x, w2t, b2)

        def rw32():
            lib.manet_tune_set(8, 2)
            return ops.conv1x1_mfma(x, w2t, b2)

        forms = {"resident weights": rw, "resident weights, 32-channel stages": rw32, "LDS weights": lds, "framework GEMM": lambda: torch.nn.functional.conv2d(x, wconv, b2),
                 "split-bf16": lambda: ops.conv1x1_split(x, sw, b2)}
        best = {k: 1e9 for k in forms}
        mean_us(rw, 200)  # warm-up
        for _ in range(4):
            for k, fn in forms.items():
                best[k] = min(best[k], mean_us(fn))
        lib.manet_tune_set(8, -2 ** 31)
        print("[%d,%d,%d,%d]: " % (B, cin, h, w) + "; ".join(
            "%s %.1f us%s" % (k, t, " = %.1f TFLOP/s (%.2f of 157.3)" % (flop / t / 1e6, flop / t / 1e6 / 157.3) if "weights" in k else "")
            for k, t in best.items()))


if __name__ == "__main__":
    main()
