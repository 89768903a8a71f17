"""Synthetic clips for the distribution-dependent measurements (bench.py `robustness`, tests/test_robustness.py).

The headline workload of bench.py draws every embedding i.i.d. (`relu(randn) * 0.1`) with uniformly random
bank labels.  That is the EASY case for anything that exploits the minimum of IntVOS.py:81-85 (the bf16 filter of
compute="bf16r") and says nothing about how plain bf16's error behaves on correlated rows.  What an encoder produces is
spatially smooth, temporally redundant and label-coherent (test.py:142-154 extracts a clip's embeddings; test.py:229-230
`rough_ROI` hands the matcher blob-shaped labels).  Three kinds, all [F, C, H, W] float32, post-ReLU, times `scale`:

  iid     relu(randn) per element, labels uniform over the ids            -- best case (bench.py's headline distribution)
  video   a smooth field (bilinear from a 1/4-resolution grid: what a stride-16 ASPP output upsampled to stride 4 looks
          like) + per-pixel detail (the stride-4 low-level branch) + an object-specific cluster offset inside each
          object's blob; frame t is the SAME scene shifted by `motion` pixels per frame with a little fresh noise (bank
          frames = temporally adjacent perturbations of the query frame); labels = the blobs    -- typical
  smooth  bilinear fields from a 1/32-resolution grid, no detail, frames nearly identical: rows indistinguishable over
          32-pixel patches and across frames; labels = blobs                -- hard case (whole 32 x 32 blocks of the bf16
          filter qualify: listed as dense entries since r4; through r4's first captures every query tile of bf16r was
          rescued by the exact fp32 kernel)
  flat    every pixel of an object carries the object's vector + 1e-4 of jitter, in every frame: NOTHING distinguishes the
          rows of an object                                                  -- the floor (more dense blocks per bucket than
          the re-rank takes: every tile is rescued by the exact fp32 kernel, and the adaptive policy then skips the filter)
"""
import math

import torch
import torch.nn.functional as F

KINDS = ("iid", "video", "smooth", "flat")


def _blob_labels(n_frames, H, W, n_ids, motion, device):
    """one ellipse per object id 1..n_ids-1, drifting `motion` pixels per frame; background = id 0.  int32 [F, H, W]"""
    yy = torch.arange(H, device=device, dtype=torch.float32)[:, None]
    xx = torch.arange(W, device=device, dtype=torch.float32)[None, :]
    labs = torch.zeros(n_frames, H, W, dtype=torch.int32, device=device)
    for o in range(1, n_ids):
        cy0 = H * (0.25 + 0.5 * ((o * 37) % 100) / 100.0)
        cx0 = W * (0.2 + 0.6 * ((o * 61) % 100) / 100.0)
        ry, rx = H * (0.12 + 0.05 * (o % 3)), W * (0.10 + 0.04 * (o % 4))
        ang = 2.0 * math.pi * ((o * 29) % 100) / 100.0
        for t in range(n_frames):
            cy, cx = cy0 + motion * t * math.sin(ang), cx0 + motion * t * math.cos(ang)
            inside = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
            labs[t][inside] = o
    return labs


def make_clip(kind, n_frames, C, H, W, n_ids, scale=0.1, device="cpu", seed=0, motion=1.5):
    """-> (embeddings [F, C, H, W] float32, labels [F, H, W] int32)"""
    if kind not in KINDS:
        raise ValueError("kind must be one of %s" % (KINDS,))
    g = torch.Generator(device=device).manual_seed(int(seed))

    def randn(*shape):
        return torch.randn(*shape, generator=g, device=device)

    if kind == "iid":
        emb = torch.relu(randn(n_frames, C, H, W)) * scale
        lab = torch.randint(0, n_ids, (n_frames, H, W), generator=g, device=device, dtype=torch.int32)
        return emb, lab
    lab = _blob_labels(n_frames, H, W, n_ids, motion, device)
    cluster = randn(n_ids, C) * 0.6  # object-specific feature offset (what makes the labels embedding-coherent)
    if kind == "flat":
        frames = [torch.relu(cluster[lab[t].long()].permute(2, 0, 1) + 0.5 + 1e-4 * randn(C, H, W)) * scale for t in range(n_frames)]
        return torch.stack(frames).contiguous(), lab
    if kind == "video":
        coarse, detail_amp, fresh_amp, temporal = 4, 0.45, 0.10, 0.04
    else:  # smooth
        coarse, detail_amp, fresh_amp, temporal = 32, 0.0, 0.0, 0.05
    # the scene: a field larger than the frame, frame t looks at it through a window shifted by motion * t pixels
    pad = int(math.ceil(motion * n_frames)) + 2
    Hs, Ws = H + 2 * pad, W + 2 * pad
    base = randn(1, C, Hs // coarse + 2, Ws // coarse + 2)
    detail = randn(1, C, Hs, Ws) * detail_amp if detail_amp > 0 else None
    frames = []
    for t in range(n_frames):
        field = F.interpolate(base + temporal * t * randn(*base.shape), size=(Hs, Ws), mode="bilinear", align_corners=True)
        if detail is not None:
            field = field + detail
        dy, dx = int(round(motion * t * 0.6)), int(round(motion * t * 0.8))
        f = field[0, :, pad + dy:pad + dy + H, pad + dx:pad + dx + W]
        f = f + cluster[lab[t].long()].permute(2, 0, 1)
        if fresh_amp > 0:
            f = f + fresh_amp * randn(C, H, W)
        frames.append(torch.relu(f) * scale)
    return torch.stack(frames).contiguous(), lab
