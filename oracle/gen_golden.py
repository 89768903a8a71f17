#!/usr/bin/env python3
"""Generate golden vectors for the matching path by RUNNING THE REFERENCE ITSELF (CPU).

Run in the build container only (needs /root/reference):   python oracle/gen_golden.py
Writes tests/golden/*.npz (inputs + the reference's outputs).  Nothing of the reference's
source travels: the fixtures are plain arrays.

The reference module is imported unmodified with three in-process shims:
  1. a dummy ``cv2`` module (imported but unused: IntVOS.py:6, config.py:5);
  2. ``sys.argv`` set (argparse runs at import: config.py:79) and
     ``torch.cuda.is_available()`` forced True only while ``config`` is imported (config.py:82-83);
  3. ``torch.Tensor.cuda`` made the identity (unconditional ``.cuda()``: IntVOS.py:102,200).
Each flag combination needs its own interpreter (cfg is frozen at import), so this script
re-invokes itself once per variant.
"""
import os
import subprocess
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference(argv):
    import torch
    sys.modules["cv2"] = types.ModuleType("cv2")
    sys.argv = ["gen_golden"] + argv
    sys.path.insert(0, REF)
    real = torch.cuda.is_available
    torch.cuda.is_available = lambda: True
    import config  # noqa: F401
    torch.cuda.is_available = real
    torch.Tensor.cuda = lambda self, *a, **k: self
    from networks import IntVOS as R
    return torch, R


def emb(torch, gen, C, h, w, scale=0.3):
    """C-major storage, as extract_feature produces (post-ReLU, non-negative)."""
    return torch.relu(torch.randn(C, h, w, generator=gen)) * scale


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        conv[k] = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **conv)
    print("wrote", name, {k: tuple(np.shape(v)) for k, v in conv.items()})


# ----------------------------------------------------------------------------------------------
def variant_global(test_mode):
    torch, R = import_reference(["--TEST_MODE", str(test_mode)])
    assert R.cfg.TEST_MODE == test_mode
    tag = "tm1" if test_mode else "tm0"
    g = torch.Generator().manual_seed(20200614)
    C, h, w = 16, 12, 15
    ref = emb(torch, g, C, h, w)
    qry = emb(torch, g, C, h, w)
    lab = torch.randint(-1, 3, (h, w, 1), generator=g).int()  # ids 0..2 present, 3 absent
    # G1 / G2: k=1, gt_ids=3 -> 4 ids, id 3 has no pixels -> 1e20
    out, ids = R.nearest_neighbor_features_per_object(
        ref.permute(1, 2, 0), qry.permute(1, 2, 0), lab, 1, gt_ids=torch.tensor(3.), n_chunks=7)
    save("global_k1_" + tag, ref_chw=ref, qry_chw=qry, labels=lab, gt_ids=3, k=1, out=out, ids=ids)
    # G3: k=3
    out3, _ = R.nearest_neighbor_features_per_object(
        ref.permute(1, 2, 0), qry.permute(1, 2, 0), lab, 3, gt_ids=torch.tensor(3.), n_chunks=2)
    save("global_k3_" + tag, ref_chw=ref, qry_chw=qry, labels=lab, gt_ids=3, k=3, out=out3)
    # gt_ids=None -> ids derived from the labels (IntVOS.py:192-198)
    outn, idsn = R.nearest_neighbor_features_per_object(
        ref.permute(1, 2, 0), qry.permute(1, 2, 0), lab, 1, gt_ids=None, n_chunks=1)
    save("global_k1_noids_" + tag, ref_chw=ref, qry_chw=qry, labels=lab, out=outn, ids=idsn)
    # G4: stacked T=2 bank, row-major [T*h, w, C] (how a caller concatenates frames)
    ref2 = emb(torch, g, C, h, w)
    lab2 = torch.randint(-1, 3, (h, w, 1), generator=g).int()
    bank = torch.cat([ref.permute(1, 2, 0), ref2.permute(1, 2, 0)], 0).contiguous()
    blab = torch.cat([lab, lab2], 0)
    out4, _ = R.nearest_neighbor_features_per_object(bank, qry.permute(1, 2, 0), blab, 1,
                                                     gt_ids=torch.tensor(2.), n_chunks=10)
    save("global_bank2_" + tag, bank_hwc=bank, qry_chw=qry, labels=blab, gt_ids=2, k=1, out=out4)
    # G6: C=100 (the real embedding width; exercises the k-padding 100 -> 104)
    C2, h2, w2 = 100, 10, 13
    refc = emb(torch, g, C2, h2, w2, 0.1)
    qryc = emb(torch, g, C2, h2, w2, 0.1)
    labc = torch.randint(-1, 2, (h2, w2, 1), generator=g).int()
    outc, _ = R.nearest_neighbor_features_per_object(
        refc.permute(1, 2, 0), qryc.permute(1, 2, 0), labc, 1, gt_ids=torch.tensor(1.), n_chunks=3)
    # a7 on top (IntVOS.py:611-612, 620-622), evaluated with the reference's own expressions
    norm = (torch.sigmoid(outc) - 0.5) * 2
    mem = torch.rand(norm.shape, generator=g)
    merged = torch.where(norm <= mem, norm, mem)
    save("global_c100_" + tag, ref_chw=refc, qry_chw=qryc, labels=labc, gt_ids=1, k=1, out=outc,
         norm=norm, mem=mem, merged=merged)


def variant_local(downsample):
    torch, R = import_reference(["--TEST_MODE", "True", "--MODEL_LOCAL_DOWNSAMPLE", str(downsample)])
    assert R.cfg.MODEL_LOCAL_DOWNSAMPLE == downsample
    tag = "ds1" if downsample else "ds0"
    g = torch.Generator().manual_seed(20200615)
    for (C, h, w, d, nobj) in [(16, 12, 15, 2, 2), (16, 12, 15, 4, 2), (100, 10, 14, 3, 3),
                               (8, 9, 11, 1, 1)]:
        prev = emb(torch, g, C, h, w)
        cur = emb(torch, g, C, h, w)
        # labels in {-1..nobj}: int_seghead feeds scribble labels with -1 through this path
        lab = torch.randint(-1, nobj + 1, (h, w, 1), generator=g).int()
        ids = torch.arange(0, nobj + 1).int()
        # L1/L3: a8 / a9 distance volume; note argument order (query, prev) (IntVOS.py:370)
        dist = R.local_pairwise_distances2(cur.permute(1, 2, 0), prev.permute(1, 2, 0), max_distance=d)
        # L2: a11
        out = R.local_previous_frame_nearest_neighbor_features_per_object(
            prev.permute(1, 2, 0), cur.permute(1, 2, 0), lab, ids, max_distance=d)
        save("local_%s_C%d_%dx%d_d%d" % (tag, C, h, w, d), prev_chw=prev, cur_chw=cur, labels=lab,
             n_ids=nobj + 1, d=d, dist=dist, out=out)


def variant_e2e():
    """E1: int_seghead + two prop_seghead rounds with tiny heads; pins dict shapes, return arity,
    a7 aggregation across rounds and a12 local-map selection through the reference's own class."""
    torch, R = import_reference(["--TEST_MODE", "True", "--MODEL_SEMANTIC_EMBEDDING_DIM", "12",
                                 "--MODEL_HEAD_EMBEDDING_DIM", "8", "--MODEL_ASPP_OUTDIM", "6",
                                 "--MODEL_MAX_LOCAL_DISTANCE", "2"])
    import torch.nn as nn
    torch.manual_seed(20200616)

    class TinyExtractor(nn.Module):  # stands in for DeepLab: [B,3,H,W] -> [B,ASPP_OUTDIM,H/4,W/4]
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(3, 6, 3, stride=4, padding=1)
            self.cls_conv = nn.Identity()
            self.upsample4 = nn.Identity()

        def forward(self, x):
            return self.conv(x)

    model = R.IntVOS(R.cfg, TinyExtractor())
    # non-trivial BN statistics so eval-mode BN is exercised
    for m in model.modules():
        if hasattr(m, "running_mean") and m.running_mean is not None:
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    model.eval()
    F_, H, W = 4, 40, 52
    imgs = torch.randn(F_, 3, H, W)
    out = {}
    with torch.no_grad():
        embs = model.extract_feature(imgs)  # [F,12,10,13]
        _, _, eh, ew = embs.shape
        nobj = 2
        scrib = torch.full((1, 1, eh, ew), -1.0)
        scrib[0, 0, 2:4, 2:9] = 1
        scrib[0, 0, 6:8, 3:11] = 2
        scrib[0, 0, 0, :] = 0
        gmap, lmaps = {}, ({}, {})
        start = 1
        seq = "clip"
        tmp, lmaps = model.int_seghead(ref_frame_embedding=embs[start:start + 1], ref_scribble_label=scrib,
                                       prev_round_label=None, global_map_tmp_dic=gmap,
                                       local_map_dics=lmaps, interaction_num=1, seq_names=[seq],
                                       gt_ids=torch.Tensor([nobj]), frame_num=[start], first_inter=True)
        out["int_logits"] = tmp[seq].clone()
        pred = torch.argmax(nn.functional.interpolate(tmp[seq], size=(H, W), mode="bilinear",
                                                      align_corners=True), dim=1)
        out["int_pred"] = pred.clone()
        prev_label = pred.unsqueeze(0)
        prev_emb = embs[start:start + 1]
        for ii in (2, 3):
            tmp, gmap, lmaps = model.prop_seghead(
                embs[start:start + 1], prev_emb, embs[ii:ii + 1], scrib, prev_label,
                normalize_nearest_neighbor_distances=True, use_local_map=True, seq_names=[seq],
                gt_ids=torch.Tensor([nobj]), k_nearest_neighbors=1, global_map_tmp_dic=gmap,
                local_map_dics=lmaps, interaction_num=1, start_annotated_frame=start, frame_num=[ii],
                dynamic_seghead=model.dynamic_seghead)
            out["prop1_logits_%d" % ii] = tmp[seq].clone()
            pred = torch.argmax(nn.functional.interpolate(tmp[seq], size=(H, W), mode="bilinear",
                                                          align_corners=True), dim=1)
            prev_label = pred.unsqueeze(0)
            prev_emb = embs[ii:ii + 1]
        out["gmap_round1"] = gmap[seq][:F_].clone()
        # second interaction round on another frame: exercises the min-merge and a12 selection
        start2 = 2
        scrib2 = torch.full((1, 1, eh, ew), -1.0)
        scrib2[0, 0, 1:3, 5:12] = 2
        scrib2[0, 0, 8, 1:6] = 0
        scrib2[0, 0, 5, 4:7] = 1
        prev_round = out["int_pred"].float().unsqueeze(0)
        tmp, lmaps = model.int_seghead(ref_frame_embedding=embs[start2:start2 + 1],
                                       ref_scribble_label=scrib2, prev_round_label=prev_round,
                                       global_map_tmp_dic=gmap, local_map_dics=lmaps, interaction_num=2,
                                       seq_names=[seq], gt_ids=torch.Tensor([nobj]), frame_num=[start2],
                                       first_inter=False)
        out["int2_logits"] = tmp[seq].clone()
        pred = torch.argmax(nn.functional.interpolate(tmp[seq], size=(H, W), mode="bilinear",
                                                      align_corners=True), dim=1)
        prev_label = pred.unsqueeze(0)
        tmp, gmap, lmaps = model.prop_seghead(
            embs[start2:start2 + 1], embs[start2:start2 + 1], embs[3:4], scrib2, prev_label,
            normalize_nearest_neighbor_distances=True, use_local_map=True, seq_names=[seq],
            gt_ids=torch.Tensor([nobj]), k_nearest_neighbors=1, global_map_tmp_dic=gmap,
            local_map_dics=lmaps, interaction_num=2, start_annotated_frame=start2, frame_num=[3],
            dynamic_seghead=model.dynamic_seghead)
        out["prop2_logits_3"] = tmp[seq].clone()
        out["gmap_round2"] = gmap[seq][:F_].clone()
        out["lmap_tmp"] = lmaps[0][seq][:F_, :2].clone()
        out["lmap_dist"] = lmaps[1][seq][:F_, :2].clone()
        out["gmap_shape"] = np.array(gmap[seq].shape)
        out["lmap_tmp_shape"] = np.array(lmaps[0][seq].shape)
        out["lmap_dist_shape"] = np.array(lmaps[1][seq].shape)
        # forward() (IntVOS.py:556-575): [ref;prev;cur] batch, global_map_tmp_dic=None
        x3 = torch.cat([imgs[1:2], imgs[2:3], imgs[3:4]], 0)
        dic = model.forward(x3, scrib, prev_label, seq_names=[seq], gt_ids=torch.Tensor([nobj]),
                            k_nearest_neighbors=1, global_map_tmp_dic=None, local_map_dics=None,
                            interaction_num=1, start_annotated_frame=1, frame_num=[3])
        out["forward_logits"] = dic[seq].clone()
        out["forward_prev_label"] = prev_label.clone()
    sd = {("sd::" + k): v for k, v in model.state_dict().items()}
    save("e2e_tiny", imgs=imgs, embs=embs, scrib=scrib, scrib2=scrib2, nobj=nobj,
         sd_keys=np.array(sorted(model.state_dict().keys())), **out, **sd)


def variant_statedict():
    """State-dict key names / shapes of the default-size IntVOS (aliased keys)."""
    torch, R = import_reference(["--TEST_MODE", "True"])
    import torch.nn as nn

    class Stub(nn.Module):
        def forward(self, x):
            return x

    model = R.IntVOS(R.cfg, Stub())
    keys = sorted(model.state_dict().keys())
    shapes = [tuple(model.state_dict()[k].shape) for k in keys]
    save("statedict_default", keys=np.array(keys), shapes=np.array([str(s) for s in shapes]))


def variant_correlation():
    """The CUDA correlation_package cannot be built here.  Tie its contract to an
    importable reference function instead: for pad=d, K=1, max_disp=d, s1=s2=1 the un-normalised
    correlation C*corr must satisfy xs + ys_shift - 2*C*corr == local_pairwise_distances2 (a9)
    inside the image.  We store a9's output as the vector."""
    torch, R = import_reference(["--TEST_MODE", "True", "--MODEL_LOCAL_DOWNSAMPLE", "False"])
    g = torch.Generator().manual_seed(20200617)
    C, h, w, d = 12, 9, 10, 2
    a = emb(torch, g, C, h, w)
    b = emb(torch, g, C, h, w)
    dist = R.local_pairwise_distances2(a.permute(1, 2, 0), b.permute(1, 2, 0), max_distance=d)
    save("correlation_tie", in1=a, in2=b, d=d, dist=dist)


def variant_mask_step():
    """8f rank 2: the driver's mask step, executed with the reference's own calls
    (test.py:253-255: interpolate bilinear align_corners + argmax; IntVOS.py:598-599: nearest + int)."""
    import torch
    import torch.nn as nn
    g = torch.Generator().manual_seed(20200618)
    for (n_ids, h, w, H, W) in [(3, 10, 13, 40, 52), (5, 12, 21, 47, 85), (2, 30, 54, 120, 214)]:
        logits = torch.randn(1, n_ids, h, w, generator=g)
        logits[0, :, 0, 0] = 0.5  # exact tie: first index must win
        pred = nn.functional.interpolate(logits, size=(H, W), mode="bilinear", align_corners=True)
        pred = torch.argmax(pred, dim=1)
        small = torch.nn.functional.interpolate(pred.unsqueeze(0).float(), size=(h, w), mode="nearest").int()
        save("mask_step_%dx%d_to_%dx%d_n%d" % (h, w, H, W, n_ids), logits=logits, mask=pred, small=small,
             size=np.array([H, W]))


def variant_seg_head():
    """8f rank 1: the reference's own _split_separable_conv2d and DynamicSegHead (IntVOS.py:488-525) in eval
    mode on random inputs: conv1->bn1->relu1 output (what the fused HIP kernel replaces) and the full head."""
    torch, R = import_reference(["--TEST_MODE", "True", "--MODEL_SEMANTIC_EMBEDDING_DIM", "13",
                                 "--MODEL_HEAD_EMBEDDING_DIM", "24"])
    torch.manual_seed(20200619)
    blk = R._split_separable_conv2d(6, 10)
    head = R.DynamicSegHead()  # in_dim 16, embed 24
    for m in list(blk.modules()) + list(head.modules()):
        if hasattr(m, "running_mean") and m.running_mean is not None:
            m.running_mean.uniform_(-0.3, 0.3)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.2, 0.2)
    blk.eval()
    head.eval()
    x = torch.randn(2, 6, 18, 67)
    with torch.no_grad():
        half = blk.relu1(blk.bn1(blk.conv1(x)))
        full = blk(x)
        hx = torch.randn(2, 16, 11, 13)
        hout = head(hx)
    sd_b = {("blk::" + k): v for k, v in blk.state_dict().items()}
    sd_h = {("head::" + k): v for k, v in head.state_dict().items()}
    save("seg_head_tiny", x=x, half=half, full=full, hx=hx, hout=hout, eps=blk.bn1.eps, **sd_b, **sd_h)


def variant_config():
    """flag names / defaults of the reference's config.py (config.py:17-80), as a JSON fixture"""
    import json
    torch, R = import_reference([])
    d = {k: v for k, v in vars(R.cfg).items() if k != "ROOT_DIR"}
    os.makedirs(OUT, exist_ok=True)
    json.dump(d, open(os.path.join(OUT, "config_defaults.json"), "w"), indent=1, sort_keys=True)
    print("wrote config_defaults", len(d))


def variant_grad():
    """8f rank 3: gradients the reference gets from torch.autograd, in TRAINING configuration (TEST_MODE False,
    train_stage1.py:126-156): d loss / d embeddings through the global match (+ normalisation), through the
    local match, and through a whole IntVOS.forward step with tiny heads (module in train() mode)."""
    torch, R = import_reference(["--TEST_MODE", "False", "--MODEL_SEMANTIC_EMBEDDING_DIM", "12",
                                 "--MODEL_HEAD_EMBEDDING_DIM", "8", "--MODEL_ASPP_OUTDIM", "6",
                                 "--MODEL_MAX_LOCAL_DISTANCE", "2"])
    import torch.nn as nn
    g = torch.Generator().manual_seed(20200620)
    out = {}
    # ---- global match, k = 1, labels incl. -1 and an object without pixels (id 3)
    C, h, w = 16, 11, 14
    ref = emb(torch, g, C, h, w).requires_grad_(True)
    qry = emb(torch, g, C, h, w).requires_grad_(True)
    lab = torch.randint(-1, 3, (h, w, 1), generator=g).int()
    o, _ = R.nearest_neighbor_features_per_object(ref.permute(1, 2, 0), qry.permute(1, 2, 0), lab, 1,
                                                  gt_ids=torch.tensor(3.), n_chunks=3)
    wgt = torch.randn(o.shape, generator=g)
    norm = (torch.sigmoid(o) - 0.5) * 2
    g_ref, g_qry = torch.autograd.grad((norm * wgt).sum(), [ref, qry])
    out.update(g_ref_chw=ref.detach(), g_qry_chw=qry.detach(), g_labels=lab, g_weight=wgt, g_out=o.detach(),
               g_grad_ref=g_ref, g_grad_qry=g_qry)
    # raw (un-normalised) distances too
    o2, _ = R.nearest_neighbor_features_per_object(ref.permute(1, 2, 0), qry.permute(1, 2, 0), lab, 1,
                                                   gt_ids=torch.tensor(2.), n_chunks=2)
    wgt2 = torch.randn(o2.shape, generator=g)
    g_ref2, g_qry2 = torch.autograd.grad((o2 * wgt2).sum(), [ref, qry])
    out.update(g_weight_raw=wgt2, g_grad_ref_raw=g_ref2, g_grad_qry_raw=g_qry2)
    # ---- local match (downsample on, the default), several sizes incl. odd ones
    for i, (C, h, w, d, nobj) in enumerate([(16, 12, 15, 2, 2), (8, 9, 11, 1, 1), (20, 14, 18, 4, 3)]):
        prev = emb(torch, g, C, h, w).requires_grad_(True)
        cur = emb(torch, g, C, h, w).requires_grad_(True)
        lab = torch.randint(0, nobj + 1, (h, w, 1), generator=g).int()
        ids = torch.arange(0, nobj + 1).int()
        o = R.local_previous_frame_nearest_neighbor_features_per_object(prev.permute(1, 2, 0), cur.permute(1, 2, 0), lab,
                                                                      ids, max_distance=d)
        wgt = torch.randn(o.shape, generator=g)
        gp, gc = torch.autograd.grad((o * wgt).sum(), [prev, cur])
        out.update({"l%d_prev_chw" % i: prev.detach(), "l%d_cur_chw" % i: cur.detach(), "l%d_labels" % i: lab,
                    "l%d_d" % i: d, "l%d_n_ids" % i: nobj + 1, "l%d_weight" % i: wgt, "l%d_out" % i: o.detach(),
                    "l%d_grad_prev" % i: gp, "l%d_grad_cur" % i: gc})
    # ---- one training step through IntVOS.forward (train_stage1.py:126-156 shape) with tiny heads
    torch.manual_seed(20200621)

    class TinyExtractor(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(3, 6, 3, stride=4, padding=1)
            self.cls_conv = nn.Identity()
            self.upsample4 = nn.Identity()

        def forward(self, x):
            return self.conv(x)

    model = R.IntVOS(R.cfg, TinyExtractor())
    model.train()
    H, W = 40, 52
    x3 = torch.randn(3, 3, H, W)  # [ref; prev; cur]
    nobj = 2
    ref_lab = torch.randint(0, nobj + 1, (1, 1, H, W)).float()
    prev_lab = torch.randint(0, nobj + 1, (1, 1, H, W)).float()
    sd0 = {("sd::" + k): v.clone() for k, v in model.state_dict().items()}
    dic = model.forward(x3, ref_lab, prev_lab, seq_names=["clip"], gt_ids=torch.Tensor([nobj]), k_nearest_neighbors=1,
                        global_map_tmp_dic=None, local_map_dics=None, interaction_num=1, start_annotated_frame=0,
                        frame_num=[2])
    logits = dic["clip"]
    wl = torch.randn(logits.shape)
    (logits * wl).sum().backward()
    names = ["feature_extracter.conv.weight", "embedding_conv.weight", "seperate_conv.weight",
             "dynamic_seghead.layer1.conv1.weight", "dynamic_seghead.conv.weight"]
    params = dict(model.named_parameters())
    for nme in names:
        out["t_grad::" + nme] = params[nme].grad.clone()
    out.update(t_x=x3, t_ref_lab=ref_lab, t_prev_lab=prev_lab, t_logits=logits.detach(), t_wl=wl, t_nobj=nobj,
               t_grad_names=np.array(names))
    save("grad_tiny", **out, **sd0)


def variant_grad_knn():
    """the gradients the reference's autograd gives through the k > 1 branch of the global match
    (IntVOS.py:87-94: topk -> where(valid, d, farthest real neighbour) -> mean), TRAINING configuration (TEST_MODE False).
    Labels include an object with FEWER than k pixels (the padding rule carries gradient) and one with none (all padding:
    `dists * valid` kills the gradient)."""
    torch, R = import_reference(["--TEST_MODE", "False"])
    g = torch.Generator().manual_seed(20200701)
    out = {}
    for i, (C, h, w, k, n_obj) in enumerate([(16, 9, 11, 3, 4), (100, 6, 7, 5, 2), (8, 5, 6, 2, 3)]):
        ref = emb(torch, g, C, h, w).requires_grad_(True)
        qry = emb(torch, g, C, h + 1, w + 2).requires_grad_(True)  # query grid differs from the bank's
        lab = torch.randint(0, 2, (h, w, 1), generator=g).int()   # ids 0 / 1 plentiful
        lab[0, 0, 0], lab[1, 2, 0] = 2, 2                         # id 2: two pixels only (< k for k = 3, 5; = k for k = 2)
        if n_obj >= 4:
            lab[2, 3, 0] = 3                                      # id 3: one pixel; id 4 (n_obj = 4): none
        o, ids = R.nearest_neighbor_features_per_object(ref.permute(1, 2, 0), qry.permute(1, 2, 0), lab, k,
                                                        gt_ids=torch.tensor(float(n_obj)), n_chunks=3)
        wgt = torch.randn(o.shape, generator=g)
        norm = (torch.sigmoid(o) - 0.5) * 2
        g_ref, g_qry = torch.autograd.grad((norm * wgt).sum(), [ref, qry])
        out.update({"c%d_ref_chw" % i: ref.detach(), "c%d_qry_chw" % i: qry.detach(), "c%d_labels" % i: lab, "c%d_k" % i: k,
                    "c%d_n_obj" % i: n_obj, "c%d_weight" % i: wgt, "c%d_out" % i: o.detach(), "c%d_grad_ref" % i: g_ref,
                    "c%d_grad_qry" % i: g_qry})
    out["n_cases"] = 3
    save("grad_knn", **out)


def variant_grad_ds0():
    """gradients of the local match with MODEL_LOCAL_DOWNSAMPLE False (IntVOS.py:299-313: raw full-resolution
    distances; :398-432: labels gathered at stride 2, constant 1.0), from the reference's own autograd.  Embeddings scaled so
    that part of the window distances fall below the constant 1.0 (only those carry gradient)."""
    torch, R = import_reference(["--TEST_MODE", "False", "--MODEL_LOCAL_DOWNSAMPLE", "False"])
    assert R.cfg.MODEL_LOCAL_DOWNSAMPLE is False
    g = torch.Generator().manual_seed(20200702)
    out = {}
    for i, (C, h, w, d, nobj, scale) in enumerate([(16, 12, 15, 2, 2, 0.45), (8, 9, 11, 1, 1, 0.7), (20, 10, 13, 3, 3, 0.4)]):
        prev = emb(torch, g, C, h, w, scale).requires_grad_(True)
        cur = emb(torch, g, C, h, w, scale).requires_grad_(True)
        lab = torch.randint(0, nobj + 1, (h, w, 1), generator=g).int()
        ids = torch.arange(0, nobj + 1).int()
        o = R.local_previous_frame_nearest_neighbor_features_per_object(prev.permute(1, 2, 0), cur.permute(1, 2, 0), lab, ids,
                                                                      max_distance=d)
        wgt = torch.randn(o.shape, generator=g)
        gp, gc = torch.autograd.grad((o * wgt).sum(), [prev, cur])
        frac = float((o < 1.0).float().mean())
        assert 0.2 < frac < 0.98, frac  # both branches of the min are exercised
        out.update({"l%d_prev_chw" % i: prev.detach(), "l%d_cur_chw" % i: cur.detach(), "l%d_labels" % i: lab, "l%d_d" % i: d,
                    "l%d_n_ids" % i: nobj + 1, "l%d_weight" % i: wgt, "l%d_out" % i: o.detach(), "l%d_grad_prev" % i: gp,
                    "l%d_grad_cur" % i: gc})
    out["n_cases"] = 3
    save("grad_ds0", **out)


def variant_grad_step_alt():
    """One training step through IntVOS.forward (train_stage1.py:126-156 shape, tiny heads, train() mode) in the configuration
    OUTSIDE the defaults that round 5 added to the HIP path: k_nearest_neighbors = 3 and MODEL_LOCAL_DOWNSAMPLE False --
    logits and parameter gradients of the reference's own run."""
    torch, R = import_reference(["--TEST_MODE", "False", "--MODEL_SEMANTIC_EMBEDDING_DIM", "12",
                                 "--MODEL_HEAD_EMBEDDING_DIM", "8", "--MODEL_ASPP_OUTDIM", "6",
                                 "--MODEL_MAX_LOCAL_DISTANCE", "2", "--MODEL_LOCAL_DOWNSAMPLE", "False"])
    import torch.nn as nn
    torch.manual_seed(20200703)

    class TinyExtractor(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(3, 6, 3, stride=4, padding=1)
            self.cls_conv = nn.Identity()
            self.upsample4 = nn.Identity()

        def forward(self, x):
            return self.conv(x)

    model = R.IntVOS(R.cfg, TinyExtractor())
    model.train()
    H, W = 40, 52
    x3 = torch.randn(3, 3, H, W) * 0.5  # [ref; prev; cur] (small inputs: part of the raw local distances stay below 1.0)
    nobj = 2
    ref_lab = torch.randint(0, nobj + 1, (1, 1, H, W)).float()
    prev_lab = torch.randint(0, nobj + 1, (1, 1, H, W)).float()
    sd0 = {("sd::" + k): v.clone() for k, v in model.state_dict().items()}
    dic = model.forward(x3, ref_lab, prev_lab, seq_names=["clip"], gt_ids=torch.Tensor([nobj]), k_nearest_neighbors=3,
                        global_map_tmp_dic=None, local_map_dics=None, interaction_num=1, start_annotated_frame=0,
                        frame_num=[2])
    logits = dic["clip"]
    wl = torch.randn(logits.shape)
    (logits * wl).sum().backward()
    names = ["feature_extracter.conv.weight", "embedding_conv.weight", "seperate_conv.weight",
             "dynamic_seghead.layer1.conv1.weight", "dynamic_seghead.conv.weight"]
    params = dict(model.named_parameters())
    out = {}
    for nme in names:
        out["t_grad::" + nme] = params[nme].grad.clone()
    out.update(t_x=x3, t_ref_lab=ref_lab, t_prev_lab=prev_lab, t_logits=logits.detach(), t_wl=wl, t_nobj=nobj, t_knn=3,
               t_grad_names=np.array(names))
    save("grad_step_alt", **out, **sd0)


def variant_rough_roi():
    """The caller-side labelling rule of the first interaction round (test.py:229-230 -> rough_ROI, test.py:323-343): what the
    bank `prop_seghead` matches against really holds.  test.py itself cannot be imported here (davisinteractive, cv2, ... at
    module level), so ONLY that function's definition is taken out of the file's syntax tree and executed as it stands, on
    seeded scribble layouts: inputs + the reference function's outputs go into the fixture, nothing of its text does."""
    import ast
    import torch
    tree = ast.parse(open(os.path.join(REF, "test.py")).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "rough_ROI"]
    assert len(fn) == 1
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fn, type_ignores=[]), os.path.join(REF, "test.py"), "exec"), ns)
    ref_fn = ns["rough_ROI"]
    out = {}
    g = torch.Generator().manual_seed(20200630)

    def layout(h, w, strokes):
        lab = torch.full((1, 1, h, w), -1.0)
        for (y0, y1, x0, x1, v) in strokes:
            lab[0, 0, y0:y1, x0:x1] = v
        return lab

    cases = [
        # strokes in the middle: most of the frame ends up background
        layout(120, 214, [(50, 53, 80, 130, 1), (70, 72, 90, 100, 2), (45, 47, 60, 70, 0)]),
        # strokes touching the top-left corner and the last row / column (the clamps max(.,0), min(.,h-1), min(.,w-1))
        layout(60, 107, [(0, 2, 0, 30, 1), (58, 60, 100, 107, 0)]),
        # one pixel; an odd grid
        layout(37, 41, [(18, 19, 20, 21, 3)]),
        # box reaching exactly the far edges minus the margin
        layout(48, 64, [(10, 28, 12, 44, 2), (5, 6, 5, 6, 0)]),
    ]
    # a batch of two different layouts (the rule is per batch element), random strokes
    rnd = torch.full((2, 1, 40, 50), -1.0)
    for b in range(2):
        for _ in range(4):
            y0, x0 = int(torch.randint(0, 35, (1,), generator=g)), int(torch.randint(0, 40, (1,), generator=g))
            rnd[b, 0, y0:y0 + int(torch.randint(1, 5, (1,), generator=g)), x0:x0 + int(torch.randint(1, 10, (1,), generator=g))] = \
                float(torch.randint(0, 3, (1,), generator=g))
    cases.append(rnd)
    for i, lab in enumerate(cases):
        out["in%d" % i] = lab
        out["out%d" % i] = ref_fn(lab.clone())
    out["n_cases"] = len(cases)
    save("rough_roi", **out)


VARIANTS = {
    "global_tm1": lambda: variant_global(True),
    "global_tm0": lambda: variant_global(False),
    "local_ds1": lambda: variant_local(True),
    "local_ds0": lambda: variant_local(False),
    "e2e": variant_e2e,
    "statedict": variant_statedict,
    "correlation": variant_correlation,
    "mask_step": variant_mask_step,
    "seg_head": variant_seg_head,
    "config": variant_config,
    "grad": variant_grad,
    "rough_roi": variant_rough_roi,
    "grad_knn": variant_grad_knn,
    "grad_ds0": variant_grad_ds0,
    "grad_step_alt": variant_grad_step_alt,
}

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in VARIANTS:
        name = sys.argv[1]
        VARIANTS[name]()
    else:
        for name in VARIANTS:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), name])
