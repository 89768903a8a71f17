#!/usr/bin/env python3
"""A test.py-shaped propagation loop on a synthetic clip, with the drop-in IntVOS on MI355X.

Mirrors the reference driver's first interaction round (test.py:137-310): extract embeddings for
the whole clip once, run the interaction head on the annotated frame, then propagate forwards and
backwards frame by frame with `prop_seghead`, feeding each predicted mask to the next frame.
No dataset, checkpoint or DAVIS session: frames and scribbles are synthetic, weights random --
this exercises the API and measures end-to-end frames/s (matching kernels + PyTorch/MIOpen heads).

    python examples/propagate_clip.py [--frames 16] [--objects 2] [--height 480 --width 854] [--fused-mask-step]
                                      [--graph]

--graph: one propagated frame (global match against the cached PreparedBank + fused local match + head input
assembly + DynamicSegHead + mask step) is captured ONCE in a HIP graph and replayed per frame: the host issues
one graph launch (plus five small device copies into / out of the graph's static buffers) instead of ~21 kernel
launches.  The masks are checked against the eager loop's.
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
    ap.add_argument("--graph", action="store_true", help="capture a propagated frame in a HIP graph and replay it")
    ap.add_argument("--framework-gemm", action="store_true",
                    help="A/B: the heads' 1x1 convolutions on the framework's GEMM (r2)")
    ap.add_argument("--pointwise", type=str, default=None, choices=["split", "f32", "framework"],
                    help="A/B: the heads' 1x1 convolutions on the split-bf16 MFMA kernel (default), the exact fp32-MFMA kernel, "
                         "or the framework's GEMM")
    ap.add_argument("--compute", type=str, default=None, help="arithmetic of the global match (f32 | bf16 | bf16x3 | bf16r)")
    ap.add_argument("--emb-dtype", type=str, default=None, help="storage of the embeddings (f32 | bf16)")
    ap.add_argument("--prepare-clip", action="store_true",
                    help="prepare every frame's operands up front (model.prepare_clip) instead of on first use")
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs the MI355X"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = make_cfg(["--TEST_MODE", "True"])
    if args.framework_gemm or args.pointwise:
        from cvpr2020_manet_amd.networks import IntVOS as _M
        _M.MFMA_POINTWISE = False if (args.framework_gemm or args.pointwise == "framework") else args.pointwise
    model = IntVOS(cfg, StandInEncoder(cfg.MODEL_ASPP_OUTDIM), compute=args.compute, emb_dtype=args.emb_dtype).to(dev).eval()
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
        if args.prepare_clip:
            embedding_memory = model.prepare_clip(embedding_memory)
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
        if not args.graph:
            return

        # ---- the same round with the propagated frame captured in a HIP graph -------------------------
        # static buffers the graph reads / writes; slot 0 of a private global-map memory stands for "this frame"
        ref = embedding_memory[start:start + 1]
        s_prev_emb, s_cur_emb = torch.empty_like(ref), torch.empty_like(ref)
        s_prev_label = torch.zeros(1, 1, H, W, dtype=torch.int64, device=dev)
        n_ids = nobj + 1
        s_gmap = {seq: torch.ones(104, eh, ew, n_ids, 1, device=dev)}

        def frame_body():
            tmp, _ = model.prop_seghead(ref, s_prev_emb, s_cur_emb, scribble, s_prev_label,
                                        normalize_nearest_neighbor_distances=True, use_local_map=True,
                                        seq_names=[seq], gt_ids=gt, k_nearest_neighbors=cfg.KNNS,
                                        global_map_tmp_dic=s_gmap, local_map_dics=None, interaction_num=1,
                                        start_annotated_frame=start, frame_num=[0],
                                        dynamic_seghead=model.dynamic_seghead)
            return mask_step(tmp[seq])

        side = torch.cuda.Stream()
        s_cur_emb.copy_(embedding_memory[0:1])
        s_prev_emb.copy_(ref)
        with torch.cuda.stream(side):  # warm the per-stream workspaces and the bank cache outside the capture
            for _ in range(2):
                frame_body()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            s_mask = frame_body()

        def graph_round():
            gmap = {}
            tmp, _ = model.int_seghead(ref_frame_embedding=ref, ref_scribble_label=scribble, prev_round_label=None,
                                       global_map_tmp_dic=gmap, local_map_dics=({}, {}), interaction_num=1,
                                       seq_names=[seq], gt_ids=gt, frame_num=[start], first_inter=True)
            ref_label = mask_step(tmp[seq]).unsqueeze(0)
            masks = {start: ref_label}
            for order in (range(start + 1, F_), range(start - 1, -1, -1)):
                prev_label, prev_emb = ref_label, ref
                for ii in order:
                    s_cur_emb.copy_(embedding_memory[ii:ii + 1])
                    s_prev_emb.copy_(prev_emb)
                    s_prev_label.copy_(prev_label)
                    s_gmap[seq][0].copy_(gmap[seq][ii])
                    graph.replay()
                    gmap[seq][ii].copy_(s_gmap[seq][0])
                    prev_label = s_mask.clone().unsqueeze(0)
                    prev_emb = embedding_memory[ii:ii + 1]
                    masks[ii] = prev_label
            return torch.cat([masks[i][0] for i in range(F_)], 0)

        graph_round()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gfinal = graph_round()
        torch.cuda.synchronize()
        gdt = time.perf_counter() - t0
        same = bool(torch.equal(gfinal, final))
        print("HIP-graph replay of the propagated frame: %.1f ms per round, %.1f frames/s (eager %.1f); masks %s the "
              "eager loop's; host work per frame: 1 graph launch + 5 small copies instead of one launch per kernel"
              % (gdt * 1e3, (F_ - 1) / gdt, (F_ - 1) / dt, "identical to" if same else "DIFFER from"))
        assert same, "graph replay changed the masks"


if __name__ == "__main__":
    main()
