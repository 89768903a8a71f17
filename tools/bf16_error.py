#!/usr/bin/env python3
"""Measured error of the bf16-family global-match modes against the fp32 MFMA kernel (which is bit-exact against the
oracle, tests/test_gpu_global.py) at BASELINE configs[2] / [4] size, whole frame, for several embedding scales.
Run on the GPU box:  python3 tools/bf16_error.py [cfg ...]      (never put this script itself after `rocprofv3 --`)
Prints one JSON object per (cfg, scale, mode): max / mean abs error of the raw distances and of the normalised maps
(sigmoid(d)-0.5)*2, and the fraction of pixels whose arg-min object id flips."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvpr2020_manet_amd import ops  # noqa: E402

CFG = {3: (120, 214, 5, 4), 5: (180, 320, 10, 6), 2: (120, 214, 5, 2)}


def main():
    cfgs = [int(a) for a in sys.argv[1:]] or [3, 5]
    dev = torch.device("cuda:0")
    for cfg in cfgs:
        H, W, T, n_ids = CFG[cfg]
        for scale in (0.1, 0.3):
            g = torch.Generator(device=dev).manual_seed(20200614 + cfg)
            cur = torch.relu(torch.randn(100, H, W, generator=g, device=dev)) * scale
            bank = torch.relu(torch.randn(T * H * W, 100, generator=g, device=dev)) * scale
            lab = torch.randint(0, n_ids, (T * H * W,), generator=g, device=dev, dtype=torch.int32)
            q = cur.permute(1, 2, 0)
            ref = ops.global_match(bank, q, lab, n_ids, compute="f32")
            refn = (torch.sigmoid(ref) - 0.5) * 2
            for mode in [m for m in ("bf16", "bf16x3", "bf16r") if m in ops.COMPUTE]:
                got = ops.global_match(bank, q, lab, n_ids, compute=mode)
                gotn = (torch.sigmoid(got) - 0.5) * 2
                e, en = (got - ref).abs(), (gotn - refn).abs()
                flips = (got.argmin(1) != ref.argmin(1)).float().mean().item()
                print(json.dumps({"cfg": cfg, "scale": scale, "mode": mode, "raw_max": e.max().item(),
                                  "raw_mean": e.mean().item(), "raw_rel_max": (e / ref.abs().clamp_min(1e-12)).max().item(),
                                  "norm_max": en.max().item(), "norm_mean": en.mean().item(), "flip_fraction": flips,
                                  "d_min": ref.min().item(), "d_mean": ref.mean().item()}), flush=True)


if __name__ == "__main__":
    main()
