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
# --warm W (r6): W untimed launches of the 1x1 kernel first, so that the N iterations that follow -- the ones
# tools/head_pmc.sh averages (pmc_summary.py --last) -- run at warm clocks, also under the counter passes
warm = int(sys.argv[sys.argv.index("--warm") + 1]) if "--warm" in sys.argv else 0
torch.manual_seed(0)
with torch.no_grad():
    x = torch.randn(3, 256, 120, 214, device="cuda")
    wt = torch.randn(256, 1, 7, 7, device="cuda")
    sc, sh = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda")
    sw = ops.SplitWeight(torch.randn(256, 256, device="cuda") * 0.05)
    w2t = torch.randn(256, 256, device="cuda") * 0.05
    b2 = torch.randn(256, device="cuda")
    hw_ = torch.randn(1, 256, 1, 1, device="cuda") * 0.05
    hb_ = torch.randn(1, device="cuda")
    gm, lm = torch.rand(120 * 214, 3, device="cuda"), torch.rand(120 * 214, 3, device="cuda")
    lab = torch.randint(0, 3, (120, 214), dtype=torch.int32, device="cuda")
    w1o, w2o = torch.randn(3, 1, 7, 7, device="cuda") * 0.1, (torch.randn(3, 256, device="cuda") * 0.05).contiguous()
    term = torch.randn(256, 120, 214, device="cuda")
    for _ in range(warm):
        ops.conv1x1_mfma(x, w2t, b2)
    for _ in range(n):
        # (r6: the per-frame set of the exact-fp32 head -- fused layer 1, depthwise, resident-weights 1x1, layer 4 with the fused output layer)
        ops.head_layer1_object(gm, lm, lab, 3, (120, 214), w1o, None, sc[:3].contiguous(), sh[:3].contiguous(), w2o, b2, term, relu_out=True)
        ops.conv1x1_mfma(x, w2t, b2, head_weight=hw_, head_bias=hb_)
        y = ops.dwconv7x7_bn_relu(x, wt, None, scale=sc, shift=sh, relu_in=False)
        y = ops.dwconv7x7_bn_relu(x, wt, None, scale=sc, shift=sh, relu_in=True)
        z = ops.conv1x1_split(y, sw, b2)
        z2 = ops.conv1x1_mfma(y, w2t, b2)
    torch.cuda.synchronize()
print("done", n)
