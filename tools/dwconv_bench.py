#!/usr/bin/env python3
"""HIP-event time of the fused depthwise-7x7 + BN + ReLU kernel at the DynamicSegHead shapes (480p grid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cvpr2020_manet_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (B, C, relu_in) in ((3, 256, False), (3, 256, True), (1, 100, False), (3, 3, False)):
    h, w = 120, 214
    x = torch.randn(B, C, h, w, device=dev)
    wt = torch.randn(C, 1, 7, 7, device=dev)
    b = torch.randn(C, device=dev)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    with torch.no_grad():
        for _ in range(5):
            ops.dwconv7x7_bn_relu(x, wt, b, scale=sc, shift=sh, relu_in=relu_in)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.dwconv7x7_bn_relu(x, wt, b, scale=sc, shift=sh, relu_in=relu_in)
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    gb = 2 * x.numel() * 4 / 1e9
    print("[%d,%d,%d,%d]%s: %.1f us per call = %.2f TB/s of the 2 x 4 B per element" % (B, C, h, w, " relu_in" if relu_in else "", us, gb / us * 1e6 / 1e3))
