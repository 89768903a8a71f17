#!/usr/bin/env python3
"""HIP-event time of the fused depthwise-7x7 + BN + ReLU kernel at the DynamicSegHead shapes (480p / 720p grids).  Per shape: a
long warm-up (the first launches of a process run at ramping clocks), then the forms alternate four times, 100 launches each; the
minimum of a form's four means is printed (box noise is one-sided).  Forms: relu_in off / on (the input read through max(x, 0)).
The same tensors are re-used (they sit in the memory-side cache); the two small shapes (~11 us) are what a python loop around the op
costs the host, not kernel times.  Two builds of the library are compared with tools/ab_so.sh."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MANET_TUNING"] = "1"
import torch  # noqa: E402

from cvpr2020_manet_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
VARIANTS = [0]


def mean_us(fn, n=100):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (B, C, h, w) in ((3, 256, 120, 214), (2, 256, 120, 214), (1, 100, 120, 214), (3, 3, 120, 214), (6, 256, 180, 320)):
    x = torch.randn(B, C, h, w, device=dev)
    wt = torch.randn(C, 1, 7, 7, device=dev)
    b = torch.randn(C, device=dev)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    forms = [(var, ri) for var in VARIANTS for ri in (False, True)]
    best, outs = {f: 1e9 for f in forms}, {}

    def run(form):
        var, ri = form
        return ops.dwconv7x7_bn_relu(x, wt, b, scale=sc, shift=sh, relu_in=ri)

    with torch.no_grad():
        mean_us(lambda: run(forms[0]), 300)  # warm-up
        for _ in range(4):
            for f in forms:
                best[f] = min(best[f], mean_us(lambda: run(f)))
        for f in forms:
            outs[f] = run(f)
    same = all(torch.equal(outs[(VARIANTS[0], ri)], outs[(v, ri)]) for v in VARIANTS for ri in (False, True))
    gb = 2 * x.numel() * 4 / 1e9
    print("[%d,%d,%d,%d] (%.0f MB in + out): %s%s" % (B, C, h, w, gb * 1e3, "; ".join(
        "%s%s %.1f us" % ("variant %d " % v if len(VARIANTS) > 1 else "", "relu_in" if ri else "plain", best[(v, ri)]) for (v, ri) in forms),
        "" if same else "  VARIANTS DIFFER"))
