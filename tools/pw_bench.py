#!/usr/bin/env python3
"""Times ops.conv1x1_mfma at [3,256,120,214] (and layer 1's shared K=100) against the framework's GEMM.
python3 tools/pw_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvpr2020_manet_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


torch.manual_seed(0)
if "--ablate" in sys.argv:  # needs MANET_TUNING=1: 1 no stores, 2 no LDS reads / MFMA, 4 no DMA behind the first two chunks
    from cvpr2020_manet_amd import _lib
    lib = _lib.load()
    with torch.no_grad():
        x = torch.randn(3, 256, 120, 214, device="cuda")
        sw = ops.SplitWeight(torch.randn(256, 256, device="cuda") * 0.05)
        b2 = torch.randn(256, device="cuda")
        for abl in (0, 1, 3, 9, 17, 25, 7):  # +8: no MFMA, +16: no activation read / split
            assert lib.manet_tune_set(3, abl) == 0
            print("ablation %d: %.1f us" % (abl, timeit(lambda: ops.conv1x1_split(x, sw, b2))))
        lib.manet_tune_set(3, 0)
    sys.exit(0)
if "--dw" in sys.argv:  # the depthwise kernel; with MANET_TUNING=1 also its ablations: 1 one kernel row only, 2 no global loads
    from cvpr2020_manet_amd import _lib
    lib = _lib.load()
    with torch.no_grad():
        x = torch.randn(3, 256, 120, 214, device="cuda")
        wt = torch.randn(256, 1, 7, 7, device="cuda")
        sc, sh = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda")
        for abl in ((0, 1, 2, 3) if os.environ.get("MANET_TUNING") == "1" else (0,)):
            if abl:
                assert lib.manet_tune_set(3, abl) == 0
            t = timeit(lambda: ops.dwconv7x7_bn_relu(x, wt, None, scale=sc, shift=sh, relu_in=True))
            print("depthwise [3,256,120,214] ablation %d: %.1f us (%.2f TB/s over in + out)" % (abl, t, 2 * x.numel() * 4 / t / 1e6))
        if os.environ.get("MANET_TUNING") == "1":
            lib.manet_tune_set(3, 0)
    sys.exit(0)
with torch.no_grad():
    for (B, cin) in ((3, 256), (2, 256), (1, 100), (3, 3)):
        x = torch.randn(B, cin, 120, 214, device="cuda")
        w2t = torch.randn(cin, 256, device="cuda") * 0.05
        b2 = torch.randn(256, device="cuda")
        w2 = w2t.t().reshape(256, cin, 1, 1).contiguous()
        t_m = timeit(lambda: ops.conv1x1_mfma(x, w2t, b2))  # (any Cin since r4)
        sw = ops.SplitWeight(w2t)
        t_s = timeit(lambda: ops.conv1x1_split(x, sw, b2))
        ref = ops.conv1x1_mfma(x, w2t, b2)
        got = ops.conv1x1_split(x, sw, b2)
        rd = torch.nn.functional.conv2d(x.double(), w2.double(), b2.double())
        print("  split-bf16 kernel %.1f us (%.2f TB/s over in + out); max |split - f64| %.3g, max |fp32 MFMA - f64| %.3g, "
              "max |framework - f64| %.3g  (|y| max %.3g)"
              % (t_s, (x.numel() + ref.numel()) * 4 / t_s / 1e6, (got.double() - rd).abs().max().item(),
                 (ref.double() - rd).abs().max().item(),
                 (torch.nn.functional.conv2d(x, w2, b2).double() - rd).abs().max().item(), rd.abs().max().item()))
        t_f = timeit(lambda: torch.nn.functional.conv2d(x, w2, b2))
        fl = 2.0 * B * 120 * 214 * cin * 256
        print("B=%d Cin=%d: MFMA kernel %.1f us (%.1f TFLOP/s = %.2f of the fp32 matrix peak), framework %.1f us"
              % (B, cin, t_m, fl / t_m / 1e6, fl / t_m / 1e6 / 157.3, t_f))
