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
with torch.no_grad():
    for (B, cin) in ((3, 256), (2, 256), (1, 100)):
        x = torch.randn(B, cin, 120, 214, device="cuda")
        w2t = torch.randn(cin, 256, device="cuda") * 0.05
        b2 = torch.randn(256, device="cuda")
        w2 = w2t.t().reshape(256, cin, 1, 1).contiguous()
        t_m = timeit(lambda: ops.conv1x1_mfma(x, w2t, b2))
        t_f = timeit(lambda: torch.nn.functional.conv2d(x, w2, b2))
        fl = 2.0 * B * 120 * 214 * cin * 256
        print("B=%d Cin=%d: MFMA kernel %.1f us (%.1f TFLOP/s = %.2f of the fp32 matrix peak), framework %.1f us"
              % (B, cin, t_m, fl / t_m / 1e6, fl / t_m / 1e6 / 157.3, t_f))
