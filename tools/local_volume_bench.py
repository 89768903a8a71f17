#!/usr/bin/env python3
"""r6: the local match split at its label boundary -- the fused kernel (r1-r5) against phase 1 batched into stored volumes
(ops.local_volumes) + phase 2 per frame on a stored volume (ops.local_match_volume).  Device time by events around loops over
ROTATING frame pairs (every frame pair of a clip has its own volume: phase 2 always reads a volume nobody touched since it was
written), us per frame pair.

python tools/local_volume_bench.py [--height 480 --width 854] [--d 12] [--pairs 60]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cvpr2020_manet_amd import _lib  # noqa: E402

if os.environ.get("MANET_LIB_VARIANT"):  # experiments: a variant build of the library (make VAR=...)
    _lib.LIB_PATH = os.path.abspath(os.environ["MANET_LIB_VARIANT"])
from cvpr2020_manet_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--width", type=int, default=854)
ap.add_argument("--d", type=int, default=12)
ap.add_argument("--ids", type=int, default=3)
ap.add_argument("--pairs", type=int, default=60)
ap.add_argument("--C", type=int, default=100)
ap.add_argument("--reps", type=int, default=5, help="repetitions of every timed loop (1 under the profiler's counter passes)")
ap.add_argument("--labels", default="blobs", choices=["blobs", "random"],
                help="previous-frame labels: a rectangle per object over background (a mask), or i.i.d. per pixel (worst case)")
a = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)
h, w = a.height // 4, (a.width + 3) // 4
F = a.pairs // 2 + 1
embs = torch.relu(torch.randn(F, a.C, h, w, device=dev)) * 0.1
frames = ops.prepare_frames(embs, compute="f32", max_distance=a.d)
if a.labels == "random":
    labs = [torch.randint(0, a.ids, (h, w), dtype=torch.int32, device=dev) for _ in range(F)]
else:
    labs = []
    for t in range(F):
        lab = torch.zeros((h, w), dtype=torch.int32, device=dev)
        for o in range(1, a.ids):
            y0, x0 = (11 * o + 2 * t) % max(h - 40, 1), (37 * o + 3 * t) % max(w - 60, 1)
            lab[y0:y0 + 36, x0:x0 + 52] = o
        labs.append(lab)
pairs = [(t - 1, t) for t in range(1, F)] + [(t + 1, t) for t in range(F - 1)]
pairs = pairs[:a.pairs]
n = len(pairs)


def timed(fn, reps=None):
    reps = a.reps if reps is None else reps
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


out = torch.ones((h, w, a.ids), dtype=torch.float32, device=dev)


def fused():
    for (p, c) in pairs:
        ops.local_match_frames(frames[p], frames[c], labs[p], a.ids, out=out, out_is_preset=False)


vols = ops.local_volumes([frames[p] for p, _ in pairs], [frames[c] for _, c in pairs])


def phase1():
    ops.local_volumes([frames[p] for p, _ in pairs], [frames[c] for _, c in pairs], out=vols)


def phase1_single():
    for i, (p, c) in enumerate(pairs):
        ops.local_volumes([frames[p]], [frames[c]], out=vols[i:i + 1])


def phase2():
    for i, (p, c) in enumerate(pairs):
        ops.local_match_volume(vols[i], frames[c], labs[p], a.ids, out=out, out_is_preset=False)


for _ in range(3):
    fused()
torch.cuda.synchronize()
tf, t1, t1s, t2 = timed(fused) / n, timed(phase1) / n, timed(phase1_single) / n, timed(phase2) / n
print("%dx%d grid, C=%d, d=%d, %d ids (%s labels), %d frame pairs, volume %.1f MB per pair" % (h, w, a.C, a.d, a.ids, a.labels, n, vols.shape[1] * 4 / 1e6))
print("  fused kernel (+ fill for d >= 11)        %7.1f us per pair" % tf)
print("  phase 1, one batched call (%2d launches)   %7.1f us per pair" % ((n + 31) // 32, t1))
print("  phase 1, one call per pair               %7.1f us per pair" % t1s)
print("  phase 2 on a stored volume (+ fill)      %7.1f us per pair   (%.2f TB/s over the volume)" % (t2, vols.shape[1] * 4 / t2 / 1e6))
