#!/usr/bin/env python3
"""Host-side cost of one ops.* call (Python + ctypes + torch allocator), GPU work kept tiny: wall time per call over
1000 back-to-back calls with one synchronise at the end, and the cProfile top of the same loop."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cvpr2020_manet_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
C, h, w = 16, 12, 16
prev = torch.randn(C, h, w, device=dev).permute(1, 2, 0)
cur = torch.randn(C, h, w, device=dev).permute(1, 2, 0)
lab = torch.randint(0, 2, (h, w), device=dev, dtype=torch.int32)
bank = ops.PreparedBank(prev, lab, 2)
x = torch.randn(1, 8, h, w, device=dev)
wt, b = torch.randn(8, 1, 7, 7, device=dev), torch.randn(8, device=dev)
cases = {
    "local_match": lambda: ops.local_match(prev, cur, lab, 2, 4, True),
    "PreparedBank.match": lambda: bank.match(cur, normalize=True),
    "dwconv7x7_bn_relu": lambda: ops.dwconv7x7_bn_relu(x, wt, b),
}
with torch.no_grad():
    for name, fn in cases.items():
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print("%-22s %.1f us of host time per call" % (name, (t1 - t0) * 1e3))
    if len(sys.argv) > 1:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(1000):
            cases[sys.argv[1]]()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
