#!/usr/bin/env python3
"""fp32 global match on prepared banks of several sizes (labelled rows of a 480p frame or of several): the kernel's device-side
split decision (one round of long splits for small / mid-size banks, split_of_block) against the host's split count everywhere
(MANET_TUNE_ONE_ROUND = 1).  Warm GPU, forms alternating, minimum of three means; results must be the same bits."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MANET_TUNING"] = "1"
import torch  # noqa: E402

from cvpr2020_manet_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda")
C, H, W = 100, 120, 214
N = H * W


def mean_us(fn, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


torch.manual_seed(0)
fq = ops.prepare_frames(torch.relu(torch.randn(C, H, W, device=dev)) * 0.1, compute="f32")
for rows, frames in ((1040, 1), (6000, 1), (17035, 1), (25680, 1), (40000, 2), (51360, 2), (128400, 5)):
    M0 = frames * N
    bank = torch.relu(torch.randn(M0, C, device=dev)) * 0.1
    lab = torch.full((M0,), -1, dtype=torch.int32, device=dev)
    idx = torch.randperm(M0, device=dev)[:rows]
    lab[idx] = torch.randint(0, 3, (rows,), device=dev, dtype=torch.int32)
    lab[idx[:rows * 9 // 10]] = 0  # most labelled rows are background, as after rough_ROI
    pb = ops.PreparedBank(bank, lab, 3)
    best, outs = {0: 1e9, 1: 1e9}, {}

    def run(f):
        lib.manet_tune_set(10, f if f else -2 ** 31)
        return pb.match(fq)

    mean_us(lambda: run(0), 30)
    for _ in range(3):
        for f in (0, 1):
            best[f] = min(best[f], mean_us(lambda: run(f)))
    for f in (0, 1):
        outs[f] = run(f)
    lib.manet_tune_set(10, -2 ** 31)
    flop = 2.0 * N * rows * C
    print("%6d labelled rows (M0 = %d): device decision %.1f us (%.3f of 157.3 TF); host splits %.1f us (%.3f)%s"
          % (rows, M0, best[0], flop / best[0] / 1e6 / 157.3, best[1], flop / best[1] / 1e6 / 157.3,
             "" if torch.equal(outs[0], outs[1]) else "  RESULTS DIFFER"))
