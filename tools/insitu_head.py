#!/usr/bin/env python3
"""Why the head's kernels take longer inside a propagated frame than in their warm micro-benchmarks: DynamicSegHead layers 2-4 on
[2, 256, 120, 214] in a loop -- alone (--match 0) or behind a 17 035-row fp32 global match per iteration as in the end-to-end loop
(--match 1) -- for `rocprofv3 --kernel-trace --stats` (tools/insitu_head.sh prints the per-kernel averages of both)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cvpr2020_manet_amd import ops  # noqa: E402
from cvpr2020_manet_amd.networks import IntVOS as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--match", type=int, default=0)
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--objects", type=int, default=2)
ap.add_argument("--fresh", type=int, default=0, help="1: a new input tensor per iteration (nothing of it in any cache)")
a = ap.parse_args()
torch.manual_seed(0)
dev = torch.device("cuda")
C, H, W = 100, 120, 214
head = M.DynamicSegHead(in_dim=103, embed_dim=256).to(dev).eval()
xs = [torch.relu(torch.randn(a.objects, 256, H, W, device=dev)) for _ in range(8 if a.fresh else 1)]
fq = ops.prepare_frames(torch.relu(torch.randn(C, H, W, device=dev)) * 0.1, compute="f32")
rows = 17035
bank = torch.relu(torch.randn(H * W, C, device=dev)) * 0.1
lab = torch.full((H * W,), -1, dtype=torch.int32, device=dev)
idx = torch.randperm(H * W, device=dev)[:rows]
lab[idx] = torch.randint(0, a.objects + 1, (rows,), device=dev, dtype=torch.int32)
lab[idx[:rows * 9 // 10]] = 0
pb = ops.PreparedBank(bank, lab, a.objects + 1)
with torch.no_grad():
    for i in range(a.iters):
        if a.match:
            pb.match(fq)
        head._tail(xs[i % len(xs)])
torch.cuda.synchronize()
