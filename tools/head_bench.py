#!/usr/bin/env python3
"""The head's two big kernels alone at [3,256,120,214] (DynamicSegHead layers 2-4 of a 2-object frame), N calls each:
python3 tools/head_bench.py [N]      (under rocprofv3 always as `-- python3 tools/head_bench.py`, never the script itself)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvpr2020_manet_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
with torch.no_grad():
    x = torch.randn(3, 256, 120, 214, device="cuda")
    wt = torch.randn(256, 1, 7, 7, device="cuda")
    sc, sh = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda")
    sw = ops.SplitWeight(torch.randn(256, 256, device="cuda") * 0.05)
    w2t = torch.randn(256, 256, device="cuda") * 0.05
    b2 = torch.randn(256, device="cuda")
    for _ in range(n):
        y = ops.dwconv7x7_bn_relu(x, wt, None, scale=sc, shift=sh, relu_in=True)
        z = ops.conv1x1_split(y, sw, b2)
        z2 = ops.conv1x1_mfma(y, w2t, b2)
    torch.cuda.synchronize()
print("done", n)
