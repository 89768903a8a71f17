#!/usr/bin/env python3
"""Can the head's depthwise (VALU / LDS) and 1x1 (matrix pipe) kernels run BESIDE each other on a CU?  (VERDICT r5 "next" #1a.)

The resident-weights 1x1 kernel takes 234 VGPRs at two waves per SIMD: 480 of a SIMD's 512 registers -- no depthwise wave (124
VGPRs) fits beside it.  This probe measures what happens when the 1x1 launch is capped to ONE workgroup per CU (half the
register file stays free): knob 11 = pixel-range groups (128 groups x 2 channel halves = 256 workgroups), knob 12 = KiB of unused
dynamic LDS (17 KiB: a second workgroup of the kernel no longer fits a CU's 160 KiB).  For each form: the 1x1 alone, the depthwise
alone, and both issued on two streams at once (independent tensors) -- wall time of the pair against the sum of the two.

MANET_TUNING=1 python tools/head_overlap_probe.py"""
import os
import sys

os.environ.setdefault("MANET_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cvpr2020_manet_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda")
torch.manual_seed(0)
C, H, W = 256, 120, 214
NBUF = 6
INT_MIN = -2 ** 31


def knobs(groups=None, pad=None):
    _lib.check(lib.manet_tune_set(11, INT_MIN if groups is None else groups), "tune")
    _lib.check(lib.manet_tune_set(12, INT_MIN if pad is None else pad), "tune")


w2t = (torch.randn(C, 256, device=dev) * 0.05).contiguous()
b2 = torch.randn(256, device=dev)
wdw = (torch.randn(C, 1, 7, 7, device=dev) * 0.1).contiguous()
sc = torch.rand(C, device=dev) + 0.5
sh = torch.randn(C, device=dev) * 0.1


def bufs(B):
    return [torch.relu(torch.randn(B, C, H, W, device=dev)) for _ in range(NBUF)]


def rw(x):
    return ops.conv1x1_mfma(x, w2t, b2, relu_out=True)


def dw(x):
    return ops.dwconv7x7_bn_relu(x, wdw, None, scale=sc, shift=sh)


def warm(xs):
    for i in range(150):
        rw(xs[i % NBUF])
    torch.cuda.synchronize()


def alone(fn, xs, n=48, reps=4):
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(xs[i % NBUF])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def pair(xs_rw, xs_dw, n=48, dw_per_rw=2, reps=4):
    """n 1x1 launches on one stream, dw_per_rw * n depthwise launches on another, issued interleaved; returns wall us per 1x1"""
    sM, sV = torch.cuda.Stream(), torch.cuda.Stream()
    best = 1e30
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True)
        eM, eV = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        sM.wait_event(e0)
        sV.wait_event(e0)
        for i in range(n):
            with torch.cuda.stream(sM):
                rw(xs_rw[i % NBUF])
            with torch.cuda.stream(sV):
                for j in range(dw_per_rw):
                    dw(xs_dw[(i * dw_per_rw + j) % NBUF])
        eM.record(sM)
        eV.record(sV)
        torch.cuda.synchronize()
        best = min(best, max(e0.elapsed_time(eM), e0.elapsed_time(eV)) * 1e3 / n)
    return best


with torch.no_grad():
    x3, x3b = bufs(3), bufs(3)
    warm(x3)
    t_dw3 = alone(dw, x3b)
    print("depthwise alone [3,256,120,214]: %.1f us" % t_dw3)
    for name, g, p in (("shipped (256 groups, 2 workgroups per CU)", None, None), ("128 groups", 128, None),
                       ("128 groups + 17 KiB pad", 128, 17), ("256 groups + 17 KiB pad (1 per CU, two rounds)", None, 17),
                       ("64 groups", 64, None)):
        knobs(g, p)
        t = alone(rw, x3)
        tp = pair(x3, x3b, dw_per_rw=2)
        print("1x1 %-48s alone %.1f us | with 2 depthwise launches beside it: %.1f us per (1x1 + 2 dw) against %.1f serial"
              % (name, t, tp, t + 2 * t_dw3))
    # smaller batches (object groups of a pipelined head)
    for B in (2, 1):
        xb, xbb = bufs(B), bufs(B)
        knobs(None, None)
        td = alone(dw, xbb)
        for name, g, p in (("shipped", None, None), ("128 groups", 128, None), ("128 groups + pad", 128, 17)):
            knobs(g, p)
            t = alone(rw, xb)
            tp = pair(xb, xbb, dw_per_rw=1)
            print("B=%d: 1x1 %-20s alone %.1f us, depthwise alone %.1f us, pair %.1f us (serial %.1f)" % (B, name, t, td, tp, t + td))
    knobs(None, None)
