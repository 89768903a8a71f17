#!/usr/bin/env python3
"""Experiment (GPU box): compute="bf16r" on the three data kinds of tools/synth_clip.py with a VARIANT build of the library
(MANET_LIB_VARIANT=path/to/libmanet_hip.so, built with other -DMANET_REFINE_CAP / -DMANET_REFINE_LDS_LIST): ms per match,
candidate rows per pair, rescued tile fraction, bit-equality with the fp32 kernel.  Not a product path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cvpr2020_manet_amd import _lib  # noqa: E402

if os.environ.get("MANET_LIB_VARIANT"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MANET_LIB_VARIANT"])
from cvpr2020_manet_amd import ops  # noqa: E402
from tools import synth_clip  # noqa: E402

dev = torch.device("cuda", 0)
H, W, C, T, n_ids = (180, 320, 100, 10, 6) if os.environ.get("SHAPE") == "cfg5" else (120, 214, 100, 5, 2)
kinds = sys.argv[1:] or ["iid", "video", "smooth"]
for kind in kinds:
    # 2 T + 1 frames, bank = the even ones, query = an odd one in the middle (temporally adjacent to its bank frames)
    emb, labs = synth_clip.make_clip(kind, 2 * T + 1, C, H, W, n_ids, scale=float(os.environ.get("SCALE", "0.1")), device=dev, seed=7)
    bank_idx, qi = list(range(0, 2 * T, 2)), T
    bank_rows = torch.cat([emb[i].permute(1, 2, 0).reshape(-1, C) for i in bank_idx], 0).contiguous()
    bank_lab = torch.cat([labs[i].reshape(-1) for i in bank_idx], 0).int().contiguous()
    q = emb[qi].permute(1, 2, 0)
    bank_r = ops.PreparedBank(bank_rows, bank_lab, n_ids, compute="bf16r")
    bank_f = ops.PreparedBank(bank_rows, bank_lab, n_ids, compute="f32")
    want = bank_f.match(q)
    got = bank_r.match(q, adaptive=False)
    st = bank_r.refine_stats_full()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        bank_r.match(q, adaptive=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print("%-7s bf16r %.3f ms per match; rows per pair %.1f; rescued tiles %d of %d; bit-equal to fp32: %s"
          % (kind, ms, st["candidate_rows_per_pair"], st["rescued_tiles"], st["query_tiles"], bool(torch.equal(want, got))))
