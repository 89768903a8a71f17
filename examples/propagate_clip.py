#!/usr/bin/env python3
"""A test.py-shaped propagation loop on a synthetic clip, with the drop-in IntVOS on MI355X.

Mirrors the reference driver's first interaction round (test.py:137-310): extract embeddings for
the whole clip once, run the interaction head on the annotated frame, then propagate forwards and
backwards frame by frame with `prop_seghead`, feeding each predicted mask to the next frame.
No dataset, checkpoint or DAVIS session: frames and scribbles are synthetic, weights random --
this exercises the API and measures end-to-end frames/s (matching kernels + PyTorch/MIOpen heads).

    python examples/propagate_clip.py [--frames 16] [--objects 2] [--height 480 --width 854] [--fused-mask-step]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from cvpr2020_manet_amd import ops  # noqa: E402
from cvpr2020_manet_amd.config import make_cfg  # noqa: E402
from cvpr2020_manet_amd.networks.IntVOS import IntVOS  # noqa: E402


class StandInEncoder(nn.Module):
    """Any module mapping [B,3,H,W] -> [B,MODEL_ASPP_OUTDIM,H/4,W/4] works as `feature_extracter`
    (the reference passes DeepLab('resnet'), test.py:70; the encoder is out of this repo's scope)."""

    def __init__(self, out_dim):
        super().__init__()
        self.net = nn.Sequential(nn.Conv2d(3, 32, 3, stride=2, padding=1), nn.ReLU(True),
                                 nn.Conv2d(32, out_dim, 3, stride=2, padding=1))

    def forward(self, x):
        return self.net(x)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--objects", type=int, default=2)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=854)
    ap.add_argument("--fused-mask-step", action="store_true",
                    help="use ops.upsample_argmax instead of F.interpolate + argmax (test.py:253-255)")
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs the MI355X"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = make_cfg(["--TEST_MODE", "True"])
    model = IntVOS(cfg, StandInEncoder(cfg.MODEL_ASPP_OUTDIM)).to(dev).eval()
    F_, H, W, nobj = args.frames, args.height, args.width, args.objects
    seq = "synthetic"

    def mask_step(logits):
        if args.fused_mask_step:
            mask, _ = ops.upsample_argmax(logits, (H, W), want_small=False)
            return mask
        pred = nn.functional.interpolate(logits, size=(H, W), mode="bilinear", align_corners=True)
        return torch.argmax(pred, dim=1)

    with torch.no_grad():
        imgs = torch.randn(F_, 3, H, W, device=dev)
        embedding_memory = torch.cat([model.extract_feature(imgs[i:i + 4]) for i in range(0, F_, 4)], 0)
        _, _, eh, ew = embedding_memory.shape
        start = F_ // 2
        scribble = torch.full((1, 1, eh, ew), -1.0, device=dev)  # -1 = unlabelled
        scribble[0, 0, 5:9, 10:60] = 0
        for o in range(1, nobj + 1):
            scribble[0, 0, 20 * o:20 * o + 6, 30 * o:30 * o + 70] = o
        gt = torch.Tensor([nobj])

        def one_round():
            gmap, lmaps = {}, ({}, {})
            ref = embedding_memory[start:start + 1]
            tmp, lmaps = model.int_seghead(ref_frame_embedding=ref, ref_scribble_label=scribble, prev_round_label=None,
                                           global_map_tmp_dic=gmap, local_map_dics=lmaps, interaction_num=1,
                                           seq_names=[seq], gt_ids=gt, frame_num=[start], first_inter=True)
            ref_label = mask_step(tmp[seq]).unsqueeze(0)
            masks = {start: ref_label}
            for order in (range(start + 1, F_), range(start - 1, -1, -1)):
                prev_label, prev_emb = ref_label, ref
                for ii in order:
                    cur = embedding_memory[ii:ii + 1]
                    tmp, gmap, lmaps = model.prop_seghead(ref, prev_emb, cur, scribble, prev_label,
                                                          normalize_nearest_neighbor_distances=True,
                                                          use_local_map=True, seq_names=[seq], gt_ids=gt,
                                                          k_nearest_neighbors=cfg.KNNS, global_map_tmp_dic=gmap,
                                                          local_map_dics=lmaps, interaction_num=1,
                                                          start_annotated_frame=start, frame_num=[ii],
                                                          dynamic_seghead=model.dynamic_seghead)
                    prev_label = mask_step(tmp[seq]).unsqueeze(0)
                    prev_emb = cur
                    masks[ii] = prev_label
            return torch.cat([masks[i][0] for i in range(F_)], 0)

        one_round()  # warm-up (MIOpen find, workspace growth)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        final = one_round()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("clip of %d frames at %dx%d (grid %dx%d), %d objects: %.1f ms per interaction round, %.1f frames/s "
          "end to end (matching + heads + mask step); masks %s"
          % (F_, H, W, eh, ew, nobj, dt * 1e3, (F_ - 1) / dt, tuple(final.shape)))


if __name__ == "__main__":
    main()
