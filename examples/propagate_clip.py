#!/usr/bin/env python3
"""A test.py-shaped propagation loop on a synthetic clip, with the drop-in IntVOS on MI355X.

Mirrors the reference driver's first interaction round (test.py:137-310): extract embeddings for
the whole clip once, run the interaction head on the annotated frame, then propagate forwards and
backwards frame by frame with `prop_seghead`, feeding each predicted mask to the next frame.
No dataset, checkpoint or DAVIS session: frames and scribbles are synthetic, weights random --
this exercises the API and measures end-to-end frames/s (matching kernels + heads + mask step; the
stand-in encoder runs before the timed region, as test.py:143-154 extracts a clip's embeddings up front).

    python examples/propagate_clip.py [--frames 16] [--objects 2] [--height 480 --width 854] [--fused-mask-step]
                                      [--graph] [--pointwise f32|split|framework] [--gpus N] [--json]

--graph: one propagated frame (global match against the cached PreparedBank + fused local match + head input
assembly + DynamicSegHead + mask step) is captured ONCE in a HIP graph and replayed per frame: the host issues
one graph launch (plus five small device copies into / out of the graph's static buffers) instead of ~21 kernel
launches.  The masks are checked against the eager loop's.

--gpus N (clip-parallel propagation, one process per GPU): what parallelises in test.py:237-259 is the
label-INDEPENDENT half of a propagated frame -- the global match against the annotated frame (4.7 ms of a 5.3 ms
frame at a 5-frame fp32 bank) -- while local match -> head -> argmax -> next frame's previous mask is a sequential
chain.  So every rank extracts the embeddings of its contiguous frame block (ONE all-gather assembles the clip, once
per clip), computes the normalised + merged global maps of its block (`IntVOS.global_maps`), ONE gather per round
ships them ([h*w*n_ids] floats per frame) to the chain rank(s), which run the chain with
`prop_seghead(..., global_map_precomputed=...)`.  The chain itself has TWO independent halves -- forwards and backwards
from the annotated frame (test.py:237-259 and :276-295) -- so with two or more ranks rank 0 runs the forward half and
rank 1 the backward half at the same time (each also runs the annotated frame's interaction head), and rank 1 ships
its masks to rank 0: up to 2x on the sequential part, whatever the number of ranks.  Rank 0 also runs the plain 1-rank
loop on the same embeddings and asserts the masks are bit-equal.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL between the ranks of a node)
# HIP streams share 4 hardware queues by default (round-robin): two streams may land on one queue and then run in order; with 8,
# every pair of streams this script creates runs side by side (--two-streams: +21 % or nothing, depending on the pair, with 4)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402

SEQ = "synthetic"


class StandInEncoder(nn.Module):
    """Any module mapping [B,3,H,W] -> [B,MODEL_ASPP_OUTDIM,H/4,W/4] works as `feature_extracter`
    (the reference passes DeepLab('resnet'), test.py:70; the encoder is out of this repo's scope)."""

    def __init__(self, out_dim):
        super().__init__()
        self.net = nn.Sequential(nn.Conv2d(3, 32, 3, stride=2, padding=1), nn.ReLU(True),
                                 nn.Conv2d(32, out_dim, 3, stride=2, padding=1))

    def forward(self, x):
        return self.net(x)


def build_model(dev, compute=None, emb_dtype=None, pointwise=None, seed=0):
    from cvpr2020_manet_amd.config import make_cfg
    from cvpr2020_manet_amd.networks.IntVOS import IntVOS
    torch.manual_seed(seed)
    cfg = make_cfg(["--TEST_MODE", "True"])
    model = IntVOS(cfg, StandInEncoder(cfg.MODEL_ASPP_OUTDIM), compute=compute, emb_dtype=emb_dtype,
                   pointwise=pointwise).to(dev).eval()
    return cfg, model


def synthetic_clip(model, dev, n_frames, H, W, nobj, frames=None, seed=1, batch=14, packed=False):
    """embeddings of the clip's frames `frames` (default: all) -- images drawn from a per-frame seed, so that every rank
    of a clip-parallel run produces the same frame.  Extracted in batches as test.py:143-154 does (batch 14); packed=True: the
    embedding layer's fused epilogue (ops.embed_finish) also leaves every frame's operands in the model's cache -- the
    batches then are NOT concatenated (a cat would copy them to storage the cache does not know): a list-backed view is
    returned instead."""
    frames = list(range(n_frames) if frames is None else frames)
    embs = []
    with torch.no_grad():
        for i0 in range(0, len(frames), batch):
            imgs = []
            for i in frames[i0:i0 + batch]:
                g = torch.Generator(device=dev).manual_seed(1000 * seed + i)
                imgs.append(torch.randn(1, 3, H, W, generator=g, device=dev))
            embs.append(model.extract_feature(torch.cat(imgs, 0), packed=packed))
    if not embs:
        return None
    if packed and len(embs) > 1:
        return BatchedClip(embs)
    return embs[0] if len(embs) == 1 else torch.cat(embs, 0)


class BatchedClip:
    """[F, C, h, w] embeddings kept as the extraction batches they were produced in (frame i = batches[i // B][i % B]): indexing
    and one-frame slices return views of the batch tensors, whose identity the model's frame cache is keyed on"""

    def __init__(self, batches):
        self.batches = batches
        self.B = batches[0].shape[0]
        self.shape = (sum(b.shape[0] for b in batches),) + tuple(batches[0].shape[1:])
        self.device, self.dtype = batches[0].device, batches[0].dtype

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            start, stop, step = idx.indices(self.shape[0])
            assert step == 1 and stop - start == 1, "one-frame slices only"
            return self.batches[start // self.B][start % self.B:start % self.B + 1]
        return self.batches[idx // self.B][idx % self.B]


def make_scribble(dev, eh, ew, nobj):
    scribble = torch.full((1, 1, eh, ew), -1.0, device=dev)  # -1 = unlabelled
    scribble[0, 0, 5:9, 10:60] = 0
    for o in range(1, nobj + 1):
        scribble[0, 0, 20 * o:20 * o + 6, 30 * o:30 * o + 70] = o
    return scribble


ROI_MARGIN = 20  # grid pixels (test.py:325)


def rough_ROI(ref_scribble_labels):
    """The first interaction round's labelling rule of the reference driver (test.py:229-230 -> test.py:323-343): inside the
    scribbles' bounding box grown by 20 grid pixels the labels stay as they are (-1 = unlabelled), EVERYTHING outside becomes
    background (0).  So the memory bank `prop_seghead` matches against in round 1 is the scribble strokes PLUS every pixel
    outside the box -- thousands of rows, not the ~1 000 pixels of the strokes.  [b, 1, h, w] -> [b, 1, h, w]; the box's far
    side is rows [.., min(h_max + 20, h - 1)) -- exclusive, so the last row / column of the grid is always outside (kept: the
    reference's slice).  Computed on the device (row / column occupancy -> two comparisons), one host read for the
    reference's error: a frame without any labelled pixel raises, as torch.min over an empty index list does there.
    Pinned by tests/golden/rough_roi.npz (the reference function's own outputs)."""
    lab = ref_scribble_labels
    b, _, h, w = lab.shape
    marked = (lab != -1).reshape(b, h, w)
    if not bool(marked.flatten(1).any(1).all()):
        raise RuntimeError("rough_ROI: a frame without labelled pixels (the reference's torch.min over an empty index list raises)")
    rows, cols = marked.any(2), marked.any(1)  # [b, h], [b, w]
    ih = torch.arange(h, device=lab.device).expand(b, h)
    iw = torch.arange(w, device=lab.device).expand(b, w)
    h_min = torch.where(rows, ih, h).amin(1, keepdim=True)
    h_max = torch.where(rows, ih, -1).amax(1, keepdim=True)
    w_min = torch.where(cols, iw, w).amin(1, keepdim=True)
    w_max = torch.where(cols, iw, -1).amax(1, keepdim=True)
    in_rows = (ih >= (h_min - ROI_MARGIN).clamp(min=0)) & (ih < (h_max + ROI_MARGIN).clamp(max=h - 1))
    in_cols = (iw >= (w_min - ROI_MARGIN).clamp(min=0)) & (iw < (w_max + ROI_MARGIN).clamp(max=w - 1))
    inside = (in_rows[:, :, None] & in_cols[:, None, :]).unsqueeze(1)
    return torch.where(inside, lab, torch.zeros_like(lab))


def full_labels(dev, eh, ew, nobj, shift=0):
    """every pixel of the grid labelled (what a propagated mask used as an annotation would give: M = h*w rows per frame, the
    bank's upper bound of SURVEY 8): one rectangle per object over background"""
    lab = torch.zeros((1, 1, eh, ew), device=dev)
    for o in range(1, nobj + 1):
        y0 = (15 * o + 3 * shift) % max(eh - 50, 1)
        x0 = (40 * o + 5 * shift) % max(ew - 80, 1)
        lab[0, 0, y0:y0 + 45, x0:x0 + 70] = o
    return lab


class StageTimer:
    """HIP-event brackets around the ops the propagated frame is made of (one instrumented round, outside any timed
    region): per-stage microseconds per frame, each stage = the launches of one ops.* call (named by its dominant kernel)."""
    STAGES = {"dwconv7x7_bn_relu": "head: dwconv7x7_bn_relu_kernel", "conv1x1_split": "head: conv1x1_x3_kernel",
              "conv1x1_mfma": "head: conv1x1_mfma_kernel", "relu_conv1x1_c1": "head: relu_conv1x1_c1_kernel",
              "local_match_frames": "local match: local_fused_kernel", "prepare_frames": "frame_prepare_kernel",
              "local_match_volume": "local match on a stored volume: local_fused_kernel<D, 2>",
              "head_inputs": "head_inputs_kernel", "head_layer1_object": "head: head_layer1_object_kernel", "upsample_argmax": "mask step: upsample_argmax_kernel",
              "label_resize_nearest": "label_resize_kernel", "frame_begin": "frame_begin_kernel (label resize + local-map pre-set + weight)"}

    def __init__(self):
        self.records = []
        self._saved = []

    def _wrap(self, obj, attr, label):
        fn = getattr(obj, attr)
        timer = self

        def wrapped(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            timer.records.append((label, e0, e1))
            return r
        self._saved.append((obj, attr, fn))
        setattr(obj, attr, wrapped)

    def __enter__(self):
        from cvpr2020_manet_amd import ops
        for name, label in self.STAGES.items():
            self._wrap(ops, name, label)
        self._wrap(ops.PreparedBank, "match", "global match: main kernel + global_finish_kernel")
        return self

    def __exit__(self, *exc):
        for obj, attr, fn in reversed(self._saved):
            setattr(obj, attr, fn)
        self._saved = []

    def per_frame_us(self, n_frames):
        torch.cuda.synchronize()
        tot, cnt = {}, {}
        for label, e0, e1 in self.records:
            tot[label] = tot.get(label, 0.0) + e0.elapsed_time(e1) * 1e3
            cnt[label] = cnt.get(label, 0) + 1
        return {k: {"us_per_frame": round(tot[k] / n_frames, 2), "calls_per_frame": round(cnt[k] / n_frames, 2)}
                for k in sorted(tot, key=lambda k: -tot[k])}


_STREAM_PAIRS = {}  # device -> the two HIP streams of Clip.one_round_two_streams


class Clip:
    """a clip's embeddings + scribble + everything one interaction round needs"""

    def __init__(self, cfg, model, embedding_memory, H, W, nobj, fused_mask_step=True, bank="roi", bank_frames=1):
        """bank: what `prop_seghead` matches against (the interaction head always sees the raw scribble, test.py:208):
             "scribble"  the strokes alone (~1 000 rows; no round of test.py produces this: r1-r4's workload)
             "roi"       the strokes after the reference driver's rough_ROI (test.py:229-230): + every pixel outside the strokes'
                         box as background -- the first interaction round's bank
             "full"      every pixel labelled (M = h*w rows per frame: the upper bound)
           bank_frames T: T annotated frames stacked along H (reference_embeddings [T*h, w, C], IntVOS.py:160-210 takes any
                         h_r; TEST_MODE passes the labels through unscaled) -- T = 5 is the metric's 5-frame memory.  The
                         annotated frames are `start` and T - 1 others spread over the clip, each with its own labels."""
        self.cfg, self.model, self.emb = cfg, model, embedding_memory
        self.F, _, self.eh, self.ew = embedding_memory.shape
        self.H, self.W, self.nobj = H, W, nobj
        self.dev = embedding_memory.device
        self.start = self.F // 2
        self.scribble = make_scribble(self.dev, self.eh, self.ew, nobj)
        self.gt = torch.Tensor([nobj])
        self.fused = fused_mask_step
        self.bank, self.bank_frames = bank, int(bank_frames)
        if bank not in ("scribble", "roi", "full"):
            raise ValueError("bank=%r (scribble | roi | full)" % (bank,))
        if not 1 <= self.bank_frames <= self.F:
            raise ValueError("bank_frames=%d for a clip of %d frames" % (self.bank_frames, self.F))
        if self.bank_frames > 1 and not cfg.TEST_MODE:
            raise ValueError("a stacked bank needs TEST_MODE (labels are passed through at grid resolution)")
        # the annotated frames of the bank: `start` first, the others spread over the clip
        others = [int(round(i * (self.F - 1) / max(self.bank_frames - 1, 1))) for i in range(self.bank_frames)]
        frames = [self.start] + [f for f in others if f != self.start]
        f = 0
        while len(frames) < self.bank_frames:
            if f not in frames:
                frames.append(f)
            f += 1
        self.bank_frame_ids = frames[:self.bank_frames]
        labs = []
        for j, _ in enumerate(self.bank_frame_ids):
            sc = self.scribble if j == 0 else torch.roll(self.scribble, shifts=(7 * j, 11 * j), dims=(2, 3))
            labs.append({"scribble": lambda: sc, "roi": lambda: rough_ROI(sc),
                         "full": lambda: full_labels(self.dev, self.eh, self.ew, nobj, shift=j)}[bank]())
        self.bank_label = labs[0] if len(labs) == 1 else torch.cat(labs, 2)  # [1, 1, T*h, w]
        if self.bank_frames == 1:
            self.bank_emb = self.emb[self.start:self.start + 1]
        else:
            self.bank_emb = torch.cat([self.emb[f:f + 1] for f in self.bank_frame_ids], 2)  # [1, C, T*h, w]
        self.bank_rows = int((self.bank_label != -1).sum().item())

    def mask_step(self, logits):
        from cvpr2020_manet_amd import ops
        if self.fused:
            mask, _ = ops.upsample_argmax(logits, (self.H, self.W), want_small=False)
            return mask
        pred = nn.functional.interpolate(logits, size=(self.H, self.W), mode="bilinear", align_corners=True)
        return torch.argmax(pred, dim=1)

    def propagation_order(self):
        return (range(self.start + 1, self.F), range(self.start - 1, -1, -1))

    def one_round(self, precomputed=None, keep_logits=None, directions=(0, 1), as_dict=False):
        """test.py:208-295 for one interaction: int_seghead on the annotated frame, then the chain.  precomputed: dict
        frame -> merged global map (IntVOS.global_maps) -- the clip-parallel form.  directions: which of the two
        independent chains to run (0 = forwards from the annotated frame, test.py:237-259; 1 = backwards, :276-295);
        as_dict: return {frame: mask} of the frames this call produced instead of the whole clip's masks"""
        model, cfg, start = self.model, self.cfg, self.start
        gmap, lmaps = {}, ({}, {})
        ref = self.emb[start:start + 1]
        tmp, lmaps = model.int_seghead(ref_frame_embedding=ref, ref_scribble_label=self.scribble, prev_round_label=None,
                                       global_map_tmp_dic=gmap, local_map_dics=lmaps, interaction_num=1,
                                       seq_names=[SEQ], gt_ids=self.gt, frame_num=[start], first_inter=True)
        ref_label = self.mask_step(tmp[SEQ]).unsqueeze(0)
        masks = {start: ref_label}
        pre = None if precomputed is None else {SEQ: precomputed}
        for di, order in enumerate(self.propagation_order()):
            if di not in directions:
                continue
            prev_label, prev_emb = ref_label, ref
            for ii in order:
                cur = self.emb[ii:ii + 1]
                tmp, gmap, lmaps = model.prop_seghead(self.bank_emb, prev_emb, cur, self.bank_label, prev_label,
                                                      normalize_nearest_neighbor_distances=True,
                                                      use_local_map=True, seq_names=[SEQ], gt_ids=self.gt,
                                                      k_nearest_neighbors=cfg.KNNS, global_map_tmp_dic=gmap,
                                                      local_map_dics=lmaps, interaction_num=1,
                                                      start_annotated_frame=start, frame_num=[ii],
                                                      dynamic_seghead=model.dynamic_seghead,
                                                      global_map_precomputed=pre)
                if keep_logits is not None:
                    keep_logits[ii] = tmp[SEQ].clone()
                prev_label = self.mask_step(tmp[SEQ]).unsqueeze(0)
                prev_emb = cur
                masks[ii] = prev_label
        if as_dict:
            return masks
        return torch.cat([masks[i][0] for i in range(self.F)], 0)

    def session(self, n_rounds, timed=True):
        """An interactive SESSION as test.py:100-310 runs it: `n_rounds` interaction rounds on one sequence with the memories
        carried over.  Round 1: scribbles on the middle frame, `rough_ROI` bank (test.py:229-230).  Round r > 1: new scribbles on
        another frame, the interaction head also sees the previous round's mask of that frame (first_inter=False,
        test.py:200-208), the bank is the new strokes ALONE (rough_ROI applies `if first_scribble` only), the global maps are
        min-merged with the earlier rounds' (IntVOS.py:615-622), the local maps compete by their distance to the annotated
        frame (:638-661).  Returns (masks of the last round [F,H,W], [seconds per round])."""
        model, cfg = self.model, self.cfg
        gmap, lmaps = {}, ({}, {})
        storage = torch.zeros((self.F, self.H, self.W), dtype=torch.int64, device=self.dev)
        times = []
        for r in range(1, n_rounds + 1):
            start = self.start if r == 1 else (self.start + (r - 1) * (self.F // 3) + 1) % self.F
            sc = self.scribble if r == 1 else torch.roll(self.scribble, shifts=(5 * r, 9 * r), dims=(2, 3))
            if timed:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            ref = self.emb[start:start + 1]
            prev_round = None if r == 1 else storage[start][None, None]
            tmp, lmaps = model.int_seghead(ref_frame_embedding=ref, ref_scribble_label=sc, prev_round_label=prev_round,
                                           global_map_tmp_dic=gmap, local_map_dics=lmaps, interaction_num=r, seq_names=[SEQ],
                                           gt_ids=self.gt, frame_num=[start], first_inter=(r == 1))
            ref_label = self.mask_step(tmp[SEQ]).unsqueeze(0)
            storage[start] = ref_label[0, 0]
            bank_label = rough_ROI(sc) if r == 1 else sc
            for order in (range(start + 1, self.F), range(start - 1, -1, -1)):
                prev_label, prev_emb = ref_label, ref
                for ii in order:
                    cur = self.emb[ii:ii + 1]
                    tmp, gmap, lmaps = model.prop_seghead(ref, prev_emb, cur, bank_label, prev_label,
                                                          normalize_nearest_neighbor_distances=True, use_local_map=True,
                                                          seq_names=[SEQ], gt_ids=self.gt, k_nearest_neighbors=cfg.KNNS,
                                                          global_map_tmp_dic=gmap, local_map_dics=lmaps, interaction_num=r,
                                                          start_annotated_frame=start, frame_num=[ii],
                                                          dynamic_seghead=model.dynamic_seghead)
                    prev_label = self.mask_step(tmp[SEQ]).unsqueeze(0)
                    prev_emb = cur
                    storage[ii] = prev_label[0, 0]
            if timed:
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
        return storage.clone(), times

    def one_round_two_streams(self):
        """The same round with the two independent halves of the chain -- forwards and backwards from the annotated frame -- on
        TWO HIP streams of this one GPU, issued alternately frame by frame: a propagated frame is ~20 launches of which a
        dozen are small (a few microseconds on a few CUs) -- while one direction runs those, the other direction's wide
        kernels have the chip.  Same kernels, same order inside a direction, disjoint frames of the memories: same masks."""
        model, cfg, start = self.model, self.cfg, self.start
        gmap, lmaps = {}, ({}, {})
        ref = self.emb[start:start + 1]
        tmp, lmaps = model.int_seghead(ref_frame_embedding=ref, ref_scribble_label=self.scribble, prev_round_label=None,
                                       global_map_tmp_dic=gmap, local_map_dics=lmaps, interaction_num=1,
                                       seq_names=[SEQ], gt_ids=self.gt, frame_num=[start], first_inter=True)
        ref_label = self.mask_step(tmp[SEQ]).unsqueeze(0)
        masks = {start: ref_label}
        n_ids = self.nobj + 1
        # the memories both directions write into (disjoint frames) exist before the streams fork
        if SEQ not in gmap:
            gmap[SEQ] = torch.ones((104, self.eh, self.ew, n_ids, 1), dtype=torch.float32, device=self.dev)
        if SEQ not in lmaps[0]:
            lmaps[0][SEQ] = torch.zeros((104, 9, self.eh, self.ew, n_ids, 1), dtype=torch.float32, device=self.dev)
        if SEQ not in lmaps[1]:
            lmaps[1][SEQ] = torch.zeros(104, 9, device=self.dev)
        # ... and so does the annotated frame's memory bank (the first propagated frame would otherwise build it on ITS stream)
        model.prepare_bank(self.bank_emb, self.bank_label, SEQ, self.gt)
        # ONE pair of streams per device and process: HIP deals a process's streams to its hardware queues round-robin, and
        # two streams on one queue run in order -- the first pair a process creates sits on two queues (measured; later
        # pairs may not: the 2nd pair with 4 queues, the 4th with 8)
        if self.dev not in _STREAM_PAIRS:
            _STREAM_PAIRS[self.dev] = (torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev))
        self._streams = _STREAM_PAIRS[self.dev]
        main = torch.cuda.current_stream(self.dev)
        orders = [list(o) for o in self.propagation_order()]
        state = [[ref_label, ref], [ref_label, ref]]
        for st_ in self._streams:
            st_.wait_stream(main)
        for i in range(max(len(o) for o in orders)):
            for di in (0, 1):
                if i >= len(orders[di]):
                    continue
                ii = orders[di][i]
                with torch.cuda.stream(self._streams[di]):
                    cur = self.emb[ii:ii + 1]
                    tmp, _, _ = model.prop_seghead(self.bank_emb, state[di][1], cur, self.bank_label, state[di][0],
                                                   normalize_nearest_neighbor_distances=True, use_local_map=True,
                                                   seq_names=[SEQ], gt_ids=self.gt, k_nearest_neighbors=cfg.KNNS,
                                                   global_map_tmp_dic=gmap, local_map_dics=lmaps, interaction_num=1,
                                                   start_annotated_frame=start, frame_num=[ii],
                                                   dynamic_seghead=model.dynamic_seghead)
                    state[di][0] = self.mask_step(tmp[SEQ]).unsqueeze(0)
                    state[di][1] = cur
                    masks[ii] = state[di][0]
        for st_ in self._streams:
            main.wait_stream(st_)
        return torch.cat([masks[i][0] for i in range(self.F)], 0)

    def timed_round(self, precomputed_fn=None, rounds=1):
        """warm-up round + `rounds` timed ones -> (masks, seconds per round)"""
        self.one_round(precomputed_fn() if precomputed_fn else None)  # warm-up (MIOpen find, workspace growth)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(rounds):
            final = self.one_round(precomputed_fn() if precomputed_fn else None)
        torch.cuda.synchronize()
        return final, (time.perf_counter() - t0) / rounds

    # ---- the same round with the propagated frame captured in a HIP graph -------------------------
    def graph_round_fn(self):
        model, cfg, start, dev = self.model, self.cfg, self.start, self.dev
        ref = self.emb[start:start + 1]
        s_prev_emb, s_cur_emb = torch.empty_like(ref), torch.empty_like(ref)
        s_prev_label = torch.zeros(1, 1, self.H, self.W, dtype=torch.int64, device=dev)
        n_ids = self.nobj + 1
        # static buffers the graph reads / writes; slot 0 of a private global-map memory stands for "this frame"
        s_gmap = {SEQ: torch.ones(104, self.eh, self.ew, n_ids, 1, device=dev)}

        def frame_body():
            tmp, _ = model.prop_seghead(self.bank_emb, s_prev_emb, s_cur_emb, self.bank_label, s_prev_label,
                                        normalize_nearest_neighbor_distances=True, use_local_map=True,
                                        seq_names=[SEQ], gt_ids=self.gt, k_nearest_neighbors=cfg.KNNS,
                                        global_map_tmp_dic=s_gmap, local_map_dics=None, interaction_num=1,
                                        start_annotated_frame=start, frame_num=[0],
                                        dynamic_seghead=model.dynamic_seghead)
            return self.mask_step(tmp[SEQ])

        side = torch.cuda.Stream()
        s_cur_emb.copy_(self.emb[0:1])
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
            tmp, _ = model.int_seghead(ref_frame_embedding=ref, ref_scribble_label=self.scribble, prev_round_label=None,
                                       global_map_tmp_dic=gmap, local_map_dics=({}, {}), interaction_num=1,
                                       seq_names=[SEQ], gt_ids=self.gt, frame_num=[start], first_inter=True)
            ref_label = self.mask_step(tmp[SEQ]).unsqueeze(0)
            masks = {start: ref_label}
            for order in self.propagation_order():
                prev_label, prev_emb = ref_label, ref
                for ii in order:
                    s_cur_emb.copy_(self.emb[ii:ii + 1])
                    s_prev_emb.copy_(prev_emb)
                    s_prev_label.copy_(prev_label)
                    s_gmap[SEQ][0].copy_(gmap[SEQ][ii])
                    graph.replay()
                    gmap[SEQ][ii].copy_(s_gmap[SEQ][0])
                    prev_label = s_mask.clone().unsqueeze(0)
                    prev_emb = self.emb[ii:ii + 1]
                    masks[ii] = prev_label
            return torch.cat([masks[i][0] for i in range(self.F)], 0)

        return graph_round


def mask_digest(masks):
    return hashlib.sha256(masks.to(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]


def run_single(args, dev, pointwise=None, want_graph=False, want_stages=False, bank=None, bank_frames=None):
    """one process, one GPU: eager (and graph) frames/s of the end-to-end propagated frame"""
    cfg, model = build_model(dev, args.compute, args.emb_dtype, pointwise if pointwise is not None else args.pointwise)
    with torch.no_grad():
        # (the producer's fused epilogue: embeddings + every frame's operands from one launch per extraction batch)
        emb = synthetic_clip(model, dev, args.frames, args.height, args.width, args.objects, packed=not args.no_packed)
        if args.prepare_clip and not isinstance(emb, BatchedClip):
            emb = model.prepare_clip(emb)
        # the label-independent half of every frame pair's local match, once per clip (model.prepare_local_volumes): the
        # window distances depend on the embeddings alone and every interaction round walks the clip again.  Timed on its own --
        # it is matching work, not encoder work: `eager_frames_per_s` is a round with the volumes (and the head's memoised
        # shared half) in place, i.e. any round but a sequence's first; `first_round_frames_per_s` charges this call to one round
        vol_ms, vol_pairs, term_frames = 0.0, 0, 0
        if not getattr(args, "no_local_volumes", False):
            # (untimed: code-object load of the batched launch, and the device allocation of the volumes -- the timed call then
            # finds its block in the caching allocator's free list, as every other timed region of this script / bench.py does)
            model.prepare_local_volumes(emb)
            model.invalidate_local_volumes()
            model.prepare_head_terms(emb)
            model.drop_head_memos()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            vol_pairs = model.prepare_local_volumes(emb)
            # ... and the propagation head's label-independent part (layer 1's shared-embedding half), 8 frames per launch
            term_frames = model.prepare_head_terms(emb)
            torch.cuda.synchronize()
            vol_ms = (time.perf_counter() - t0) * 1e3
        clip = Clip(cfg, model, emb, args.height, args.width, args.objects, fused_mask_step=args.fused_mask_step,
                    bank=bank if bank is not None else args.bank,
                    bank_frames=bank_frames if bank_frames is not None else args.bank_frames)
        final, dt = clip.timed_round(rounds=args.rounds)
        res = {"frames": args.frames, "grid": [clip.eh, clip.ew], "objects": args.objects, "pointwise": model.pointwise,
               "compute": model.compute, "bank": clip.bank, "bank_frames": clip.bank_frames, "bank_rows": clip.bank_rows,
               "eager_ms_per_round": dt * 1e3, "eager_frames_per_s": (args.frames - 1) / dt,
               "local_volumes": {"pairs": vol_pairs, "head_term_frames": term_frames, "ms": vol_ms,
                                 "MB": model.local_volume_bytes_cached() / 1e6},
               "first_round_frames_per_s": (args.frames - 1) / (dt + vol_ms * 1e-3),
               "mask_digest": mask_digest(final)}
        if want_stages:
            with StageTimer() as st:
                clip.one_round()
            res["per_frame_stages_us"] = st.per_frame_us(args.frames - 1)
        if getattr(args, "two_streams", False):
            clip.one_round_two_streams()  # warm-up: per-stream workspaces
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.rounds):
                tfinal = clip.one_round_two_streams()
            torch.cuda.synchronize()
            tdt = (time.perf_counter() - t0) / args.rounds
            res.update({"two_streams_ms_per_round": tdt * 1e3, "two_streams_frames_per_s": (args.frames - 1) / tdt,
                        "two_streams_masks_equal_eager": bool(torch.equal(tfinal, final))})
        if getattr(args, "session", 0):
            clip.session(min(args.session, 2), timed=False)  # warm-up: both kinds of round
            smask, times = clip.session(args.session)
            again, _ = clip.session(args.session, timed=False)
            # (the volumes are part of the session's matching work: their one-off cost is charged to it)
            res.update({"session_rounds": args.session, "session_ms_per_round": [t * 1e3 for t in times],
                        "session_frames_per_s": args.session * (args.frames - 1) / (sum(times) + vol_ms * 1e-3),
                        "session_frames_per_s_rounds_alone": args.session * (args.frames - 1) / sum(times),
                        "session_mask_digest": mask_digest(smask), "session_repeatable": bool(torch.equal(smask, again))})
        if want_graph:
            ground = clip.graph_round_fn()
            ground()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.rounds):
                gfinal = ground()
            torch.cuda.synchronize()
            gdt = (time.perf_counter() - t0) / args.rounds
            res.update({"graph_ms_per_round": gdt * 1e3, "graph_frames_per_s": (args.frames - 1) / gdt,
                        "graph_masks_equal_eager": bool(torch.equal(gfinal, final))})
    return res, clip, final


def run_parallel(args, dev, rank, world):
    """clip-parallel propagation (module docstring): returns rank 0's result dict (None elsewhere)"""
    from cvpr2020_manet_amd import clip_parallel as cp
    cfg, model = build_model(dev, args.compute, args.emb_dtype, args.pointwise)
    F_ = args.frames
    s0, e0 = cp.shard_frames(F_, world, rank)
    with torch.no_grad():
        # sharded feature extraction, then ONE all-gather assembles the clip on every rank (once per clip)
        mine = synthetic_clip(model, dev, F_, args.height, args.width, args.objects, frames=range(s0, e0))
        if mine is None:
            probe = synthetic_clip(model, dev, F_, args.height, args.width, args.objects, frames=[0])
            mine = probe[:0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        emb = cp.all_gather_clip(mine, F_)
        torch.cuda.synchronize()
        clip_gather_ms = (time.perf_counter() - t0) * 1e3
        clip = Clip(cfg, model, emb, args.height, args.width, args.objects, fused_mask_step=args.fused_mask_step,
                    bank=args.bank, bank_frames=args.bank_frames)
        start = clip.start
        ref = emb[start:start + 1]
        my_frames = [f for f in range(s0, e0)]
        L = clip.eh * clip.ew * (args.objects + 1)
        timing = {}

        # the two directions of the propagation (forwards / backwards from the annotated frame) are independent chains:
        # rank 0 runs the forward one, rank 1 -- when there is one -- the backward one, then ships its masks to rank 0
        chain_ranks = (0, 1) if world > 1 and start > 0 else (0,)
        # the chain's label-independent work, once per clip ON the chain rank: the window-distance volumes of the frame pairs of
        # its direction, one batched call (33 us per pair at 480p, d = 12).  Not sharded over the ranks: a volume is 25.8 MB --
        # 169 us over one 153 GB/s xGMI link, 24 us over all seven into the chain rank -- shipping it costs what computing it
        # costs, and it is needed once per clip (every later round finds it stored); the global maps it is sharded "alongside"
        # are 0.3 MB per frame.
        vol_ms = 0.0
        if rank in chain_ranks and not getattr(args, "no_local_volumes", False):
            fwd = [(t - 1, t) for t in range(start + 1, F_)]
            bwd = [(t + 1, t) for t in range(start - 1, -1, -1)]
            mine_pairs = fwd + bwd if len(chain_ranks) == 1 else (fwd if rank == 0 else bwd)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            emb_ = model.prepare_clip(emb)  # (keeps every frame's operands: the volumes and the head's terms hang on them)
            assert emb_ is emb
            model.prepare_local_volumes(emb, pairs=mine_pairs)
            model.prepare_head_terms(emb)   # the propagation head's label-independent part, too (8 frames per launch)
            torch.cuda.synchronize()
            vol_ms = (time.perf_counter() - t0) * 1e3

        def maps_for_round():
            """this rank's block -> normalised + merged global maps; ONE collective ships every rank's to the chain rank(s)"""
            t1 = time.perf_counter()
            if my_frames:
                rows = model.global_maps(clip.bank_emb, clip.bank_label, emb[s0:e0], my_frames, SEQ, clip.gt)
            else:
                rows = torch.empty((0, L), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            timing["global_maps_ms"] = (time.perf_counter() - t1) * 1e3
            allrows = cp.gather_frame_rows(rows, F_, dst=0 if len(chain_ranks) == 1 else None, timing=True)
            timing["gather"] = dict(cp.LAST_GATHER)
            if rank not in chain_ranks:
                return None
            return {f: allrows[f] for f in range(F_) if f != start}

        def one_parallel_round():
            pre = maps_for_round()
            if rank not in chain_ranks:
                return None
            # the SEQUENTIAL part, timed on its own: what no number of ranks shortens
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            if len(chain_ranks) == 1:
                out_ = clip.one_round(pre)
                torch.cuda.synchronize()
                timing["chain_ms"], timing["chain_frames"] = (time.perf_counter() - t1) * 1e3, F_ - 1
                return out_
            mine_ = clip.one_round(pre, directions=(rank,), as_dict=True)
            torch.cuda.synchronize()
            timing["chain_ms"], timing["chain_frames"] = (time.perf_counter() - t1) * 1e3, max(len(mine_) - 1, 1)
            if rank == 1:  # the backward chain's masks (frames start - 1 .. 0) -> rank 0, as int16
                back = torch.cat([mine_[i][0] for i in range(start)], 0)
                cp.send_tensor(back.to(torch.int16), dst=0)
                return None
            back = cp.recv_tensor((start, args.height, args.width), torch.int16, dev, src=1)
            fwd = torch.cat([mine_[i][0] for i in range(start, F_)], 0)
            return torch.cat([back.to(fwd.dtype), fwd], 0)

        one_parallel_round()  # warm-up
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        final = one_parallel_round()
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        if rank != 0:
            return None
        # the plain 1-rank loop on the same embeddings: the masks must be the same bits
        want, dt1 = clip.timed_round()
        same = bool(torch.equal(final, want))
        # Amdahl bookkeeping of the flow: per propagated frame, the part that shards over the ranks (the global match:
        # this rank's block, `global_maps_ms`) and the chain (local match on the stored volume + head + mask step, frame by
        # frame on the chain rank).  ceiling = the speed-up over ONE rank's loop that no number of ranks exceeds: (sharded +
        # chain) / chain, times the number of chain ranks (the two directions are independent chains).
        coll = dict(timing.get("gather") or {})
        n_mine = max(len(my_frames), 1)
        sharded_us = timing.get("global_maps_ms", 0.0) * 1e3 / n_mine
        chain_us = timing.get("chain_ms", 0.0) * 1e3 / max(timing.get("chain_frames", 1), 1)
        coll.update({"sharded_us_per_frame": sharded_us, "chain_us_per_frame": chain_us, "chain_ranks": len(chain_ranks),
                     "amdahl_ceiling": (sharded_us + chain_us) / max(chain_us, 1e-9) * len(chain_ranks),
                     "measured_speedup": dt1 / dt})
        timing["gather"] = coll
        return {"frames": F_, "world": world, "backend": dist.get_backend(), "pointwise": model.pointwise,
                "compute": model.compute, "bank": clip.bank, "bank_frames": clip.bank_frames, "bank_rows": clip.bank_rows,
                "parallel_ms_per_round": dt * 1e3, "parallel_frames_per_s": (F_ - 1) / dt,
                "single_rank_ms_per_round": dt1 * 1e3, "single_rank_frames_per_s": (F_ - 1) / dt1,
                "masks_bit_equal_to_single_rank": same, "mask_digest": mask_digest(final),
                "chain_ranks": list(chain_ranks),
                "clip_all_gather_ms": clip_gather_ms, "rank0_global_maps_ms": timing.get("global_maps_ms"),
                "rank0_local_volumes_ms": vol_ms, "rank0_local_volume_MB": model.local_volume_bytes_cached() / 1e6,
                "collective": timing.get("gather")}


def spawn_ranks(n, argv):
    """start the N ranks (one process per GPU, torch.distributed.run, rendezvous on 127.0.0.1) before anything in this
    process touches the GPU, relay the output, pass the exit code on"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--objects", type=int, default=2)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=854)
    ap.add_argument("--fused-mask-step", action="store_true",
                    help="use ops.upsample_argmax instead of F.interpolate + argmax (test.py:253-255)")
    ap.add_argument("--graph", action="store_true", help="capture a propagated frame in a HIP graph and replay it")
    ap.add_argument("--pointwise", type=str, default=None, choices=["split", "split3", "f32", "framework"],
                    help="the heads' 1x1 convolutions: exact fp32-MFMA kernel (default), the split-bf16 MFMA kernel, or the "
                         "framework's GEMM")
    ap.add_argument("--compute", type=str, default=None, help="arithmetic of the global match (f32 | bf16 | bf16x3 | bf16r)")
    ap.add_argument("--emb-dtype", type=str, default=None, help="storage of the embeddings (f32 | bf16)")
    ap.add_argument("--prepare-clip", action="store_true",
                    help="prepare every frame's operands up front (model.prepare_clip) instead of on first use")
    ap.add_argument("--no-packed", action="store_true",
                    help="extract the embeddings through the stock module chain (bn2, relu2, cast as separate passes; frames "
                         "prepared on first use) instead of the fused embedding epilogue (extract_feature(packed=True))")
    ap.add_argument("--bank", type=str, default="roi", choices=["scribble", "roi", "full"],
                    help="labels of the bank prop_seghead matches against: roi = the scribble after the reference driver's "
                         "rough_ROI (test.py:229-230: everything outside the strokes' box +-20 is background -- the default, what "
                         "test.py's first round produces); scribble = the strokes alone (r1-r4's workload); full = every pixel")
    ap.add_argument("--bank-frames", type=int, default=1,
                    help="annotated frames stacked into the bank (5 = the metric's 5-frame memory)")
    ap.add_argument("--rounds", type=int, default=1, help="timed interaction rounds (after one warm-up round)")
    ap.add_argument("--session", type=int, default=0,
                    help="also run a whole interactive session of this many rounds (test.py:100-310: round 1 on the rough_ROI "
                         "bank, later rounds on new strokes alone with the memories carried over); the reference runs 8")
    ap.add_argument("--no-local-volumes", action="store_true",
                    help="do not store the frame pairs' window-distance volumes up front (model.prepare_local_volumes): every "
                         "propagated frame then runs the fused local kernel, as r1-r5")
    ap.add_argument("--two-streams", action="store_true",
                    help="also time the round with the forward and the backward half of the chain on two HIP streams")
    ap.add_argument("--stages", action="store_true", help="per-stage microseconds of a propagated frame (HIP events)")
    ap.add_argument("--gpus", type=int, default=1, help="clip-parallel propagation over N ranks (module docstring)")
    ap.add_argument("--json", action="store_true", help="print the result as one JSON line")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    backend = os.environ.get("MANET_BENCH_BACKEND", "nccl")  # gloo: dry run of the N-rank flow on fewer GPUs than ranks
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit("propagate_clip.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    assert torch.cuda.is_available(), "needs the MI355X"
    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % n_dev)
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        res = run_parallel(args, dev, rank, world)
        dist.destroy_process_group()
        if rank == 0:
            if args.json:
                print(json.dumps(res), flush=True)
            else:
                print("clip-parallel propagation, %d ranks (%s): %.1f ms per round = %.1f frames/s (1 rank on the same "
                      "embeddings: %.1f); masks %s the 1-rank loop's; gather of the round's global maps: %s"
                      % (world, res["backend"], res["parallel_ms_per_round"], res["parallel_frames_per_s"],
                         res["single_rank_frames_per_s"],
                         "bit-equal to" if res["masks_bit_equal_to_single_rank"] else "DIFFER from", res["collective"]))
            assert res["masks_bit_equal_to_single_rank"], "clip-parallel propagation changed the masks"
        return
    res, clip, final = run_single(args, dev, want_graph=args.graph, want_stages=args.stages)
    if args.json:
        print(json.dumps(res), flush=True)
    else:
        print("clip of %d frames at %dx%d (grid %dx%d), %d objects, head 1x1 = %s: %.1f ms per interaction round, %.1f "
              "frames/s end to end (matching + heads + mask step); masks %s"
              % (args.frames, args.height, args.width, clip.eh, clip.ew, args.objects, res["pointwise"],
                 res["eager_ms_per_round"], res["eager_frames_per_s"], tuple(final.shape)))
        if args.stages:
            for k, v in res["per_frame_stages_us"].items():
                print("  %-55s %8.1f us per frame (%.1f calls)" % (k, v["us_per_frame"], v["calls_per_frame"]))
        if args.graph:
            print("HIP-graph replay of the propagated frame: %.1f ms per round, %.1f frames/s (eager %.1f); masks %s the "
                  "eager loop's; host work per frame: 1 graph launch + 5 small copies instead of one launch per kernel"
                  % (res["graph_ms_per_round"], res["graph_frames_per_s"], res["eager_frames_per_s"],
                     "identical to" if res["graph_masks_equal_eager"] else "DIFFER from"))
        if args.session:
            print("session of %d interaction rounds (round 1: rough_ROI bank, then strokes alone, memories carried over): %s ms per "
                  "round, %.1f frames/s over the session" % (args.session, ", ".join("%.1f" % t for t in res["session_ms_per_round"]),
                                                             res["session_frames_per_s"]))
        if args.two_streams:
            print("forward and backward halves of the chain on two HIP streams: %.1f ms per round, %.1f frames/s (one stream %.1f); "
                  "masks %s the one-stream loop's" % (res["two_streams_ms_per_round"], res["two_streams_frames_per_s"],
                                                      res["eager_frames_per_s"],
                                                      "identical to" if res["two_streams_masks_equal_eager"] else "DIFFER from"))
    if args.graph:
        assert res["graph_masks_equal_eager"], "graph replay changed the masks"
    if args.two_streams:
        assert res["two_streams_masks_equal_eager"], "the two-stream round changed the masks"


if __name__ == "__main__":
    main()
