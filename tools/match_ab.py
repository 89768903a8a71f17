#!/usr/bin/env python3
"""A/B of the global-match main kernel on a scribble-sized bank: query operand from the per-call pack pass vs from a
prepared frame (fresh buffer per frame).  python3 tools/match_ab.py"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvpr2020_manet_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
h, w, C, n_ids = 120, 214, 100, 3
embs = torch.relu(torch.randn(12, C, h, w, device=dev)) * 0.1
lab = torch.full((h * w,), -1, dtype=torch.int32, device=dev)
lab[600:1200] = 0
lab[5000:6500] = 1
lab[9000:9800] = 2
bank = ops.PreparedBank(embs[0].permute(1, 2, 0), lab, n_ids)


def run(tag, fn, n=10):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    _lib.check(lib.manet_profile_begin(n), "begin")
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    ms = (ctypes.c_float * n)()
    k = ctypes.c_int(0)
    _lib.check(lib.manet_profile_end(ms, n, ctypes.byref(k)), "end")
    print(tag, ["%.1f" % (ms[i] * 1e3) for i in range(k.value)])


frames = ops.prepare_frames(embs, max_distance=12)
run("per-call pack      ", lambda i: bank.match(embs[1 + i % 11].permute(1, 2, 0), normalize=True))
run("prepared (old frames)", lambda i: bank.match(frames[1 + i % 11], normalize=True))
run("prepare + match     ", lambda i: bank.match(ops.prepare_frames(embs[1 + i % 11], max_distance=12), normalize=True))
pq = [ops.PackedQuery(embs[i].permute(1, 2, 0)) for i in range(12)]
run("PackedQuery (old)    ", lambda i: bank.match(pq[1 + i % 11], normalize=True))
for (name, fill) in (("example-sized scribbles (1040 rows)", [(600, 800, 0), (5000, 5420, 1), (9000, 9420, 2)]),
                     ("one fully labelled frame", [(0, 9000, 0), (9000, 18000, 1), (18000, 25680, 2)])):
    lab2 = torch.full((h * w,), -1, dtype=torch.int32, device=dev)
    for a, b, o in fill:
        lab2[a:b] = o
    for compute in ("f32", "bf16", "bf16x3"):
        bk = ops.PreparedBank(embs[0].permute(1, 2, 0), lab2, n_ids, compute=compute)
        fr = ops.prepare_frames(embs, compute=compute)
        run("%s, %s" % (name, compute), lambda i: bk.match(fr[1 + i % 11], normalize=True), n=6)
        ref = ops.global_match(embs[0].permute(1, 2, 0), embs[3].permute(1, 2, 0), lab2, n_ids, compute=compute)
        assert torch.equal(bk.match(fr[3]), ref)
