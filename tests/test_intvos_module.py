"""The drop-in networks/IntVOS.py counterpart against the reference's own class (golden e2e_tiny.npz,
made by running the reference's IntVOS.int_seghead / prop_seghead / forward with tiny heads).

* CPU (host logic): the three ops entry points are replaced -- IN THIS TEST ONLY -- by oracle-backed
  stand-ins so that the dict plumbing, return arity, label scaling, memory updates (a7, a12) and
  head-input assembly can be checked without a GPU.  The product has no such path.
* GPU: the same script through the real HIP ops.
"""
import argparse

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_golden


class TinyExtractor(nn.Module):  # same stand-in encoder as oracle/gen_golden.py
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 6, 3, stride=4, padding=1)
        self.cls_conv = nn.Identity()
        self.upsample4 = nn.Identity()

    def forward(self, x):
        return self.conv(x)


def tiny_cfg():
    from cvpr2020_manet_amd.config import make_cfg
    return make_cfg(["--TEST_MODE", "True", "--MODEL_SEMANTIC_EMBEDDING_DIM", "12",
                     "--MODEL_HEAD_EMBEDDING_DIM", "8", "--MODEL_ASPP_OUTDIM", "6",
                     "--MODEL_MAX_LOCAL_DISTANCE", "2"])


def build_model(g, device, **kw):
    from cvpr2020_manet_amd.networks import IntVOS as M
    model = M.IntVOS(tiny_cfg(), TinyExtractor(), **kw)
    sd = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd::")}
    assert sorted(model.state_dict().keys()) == sorted(sd.keys()) == sorted(g["sd_keys"].tolist())
    model.load_state_dict(sd, strict=True)
    return model.to(device).eval()


def run_script(model, g, device):
    """the exact call sequence of oracle/gen_golden.py:variant_e2e"""
    t = lambda a: torch.from_numpy(a).to(device)
    imgs, scrib, scrib2 = t(g["imgs"]), t(g["scrib"]), t(g["scrib2"])
    nobj = int(g["nobj"])
    F_, H, W = imgs.shape[0], imgs.shape[2], imgs.shape[3]
    out = {}
    up = lambda x: torch.argmax(nn.functional.interpolate(x, size=(H, W), mode="bilinear", align_corners=True), dim=1)
    with torch.no_grad():
        embs = model.extract_feature(imgs)
        out["embs"] = embs
        gmap, lmaps, seq, start = {}, ({}, {}), "clip", 1
        gt = torch.Tensor([nobj])
        tmp, lmaps = model.int_seghead(ref_frame_embedding=embs[start:start + 1], ref_scribble_label=scrib,
                                       prev_round_label=None, global_map_tmp_dic=gmap, local_map_dics=lmaps,
                                       interaction_num=1, seq_names=[seq], gt_ids=gt, frame_num=[start],
                                       first_inter=True)
        out["int_logits"] = tmp[seq]
        pred = up(tmp[seq]); out["int_pred"] = pred
        prev_label, prev_emb = pred.unsqueeze(0), embs[start:start + 1]
        for ii in (2, 3):
            res = model.prop_seghead(embs[start:start + 1], prev_emb, embs[ii:ii + 1], scrib, prev_label,
                                     normalize_nearest_neighbor_distances=True, use_local_map=True,
                                     seq_names=[seq], gt_ids=gt, k_nearest_neighbors=1, global_map_tmp_dic=gmap,
                                     local_map_dics=lmaps, interaction_num=1, start_annotated_frame=start,
                                     frame_num=[ii], dynamic_seghead=model.dynamic_seghead)
            assert len(res) == 3  # (dic, global_map_tmp_dic, local_map_dics): IntVOS.py:681
            tmp, gmap, lmaps = res
            out["prop1_logits_%d" % ii] = tmp[seq]
            prev_label, prev_emb = up(tmp[seq]).unsqueeze(0), embs[ii:ii + 1]
        out["gmap_round1"] = gmap[seq][:F_].clone()
        start2 = 2
        prev_round = out["int_pred"].float().unsqueeze(0)
        tmp, lmaps = model.int_seghead(ref_frame_embedding=embs[start2:start2 + 1], ref_scribble_label=scrib2,
                                       prev_round_label=prev_round, global_map_tmp_dic=gmap, local_map_dics=lmaps,
                                       interaction_num=2, seq_names=[seq], gt_ids=gt, frame_num=[start2],
                                       first_inter=False)
        out["int2_logits"] = tmp[seq]
        prev_label = up(tmp[seq]).unsqueeze(0)
        tmp, gmap, lmaps = model.prop_seghead(embs[start2:start2 + 1], embs[start2:start2 + 1], embs[3:4], scrib2,
                                              prev_label, normalize_nearest_neighbor_distances=True,
                                              use_local_map=True, seq_names=[seq], gt_ids=gt,
                                              k_nearest_neighbors=1, global_map_tmp_dic=gmap, local_map_dics=lmaps,
                                              interaction_num=2, start_annotated_frame=start2, frame_num=[3],
                                              dynamic_seghead=model.dynamic_seghead)
        out["prop2_logits_3"] = tmp[seq]
        out["gmap_round2"] = gmap[seq][:F_].clone()
        out["lmap_tmp"] = lmaps[0][seq][:F_, :2].clone()
        out["lmap_dist"] = lmaps[1][seq][:F_, :2].clone()
        out["gmap_shape"] = torch.tensor(gmap[seq].shape)
        out["lmap_tmp_shape"] = torch.tensor(lmaps[0][seq].shape)
        out["lmap_dist_shape"] = torch.tensor(lmaps[1][seq].shape)
        x3 = torch.cat([imgs[1:2], imgs[2:3], imgs[3:4]], 0)
        dic = model.forward(x3, scrib, t(g["forward_prev_label"]), seq_names=[seq], gt_ids=gt,
                            k_nearest_neighbors=1, global_map_tmp_dic=None, local_map_dics=None, interaction_num=1,
                            start_annotated_frame=1, frame_num=[3])
        assert isinstance(dic, dict)  # IntVOS.py:675-676: a bare dict when there is no global memory
        out["forward_logits"] = dic[seq]
        # prop_seghead with a global memory but no local memory returns a pair (IntVOS.py:678-679)
        res = model.prop_seghead(embs[1:2], embs[2:3], embs[3:4], scrib, prev_label, seq_names=[seq], gt_ids=gt,
                                 global_map_tmp_dic={}, local_map_dics=None, interaction_num=1,
                                 start_annotated_frame=1, frame_num=[3], dynamic_seghead=model.dynamic_seghead)
        assert len(res) == 2
    return out


def compare(out, g, tol_logits):
    for k in ["gmap_shape", "lmap_tmp_shape", "lmap_dist_shape"]:
        assert out[k].tolist() == g[k].tolist(), k  # [104,h,w,n_ids,1], [104,9,h,w,n_ids,1], [104,9]
    np.testing.assert_allclose(out["embs"].cpu().numpy(), g["embs"], rtol=1e-4, atol=1e-5)
    for k in ["gmap_round1", "gmap_round2", "lmap_tmp", "lmap_dist"]:
        np.testing.assert_allclose(out[k].cpu().numpy(), g[k], rtol=1e-4, atol=2e-5, err_msg=k)
    for k in ["int_logits", "prop1_logits_2", "prop1_logits_3", "int2_logits", "prop2_logits_3", "forward_logits"]:
        np.testing.assert_allclose(out[k].cpu().numpy(), g[k], rtol=tol_logits, atol=tol_logits, err_msg=k)
    np.testing.assert_array_equal(out["int_pred"].cpu().numpy(), g["int_pred"])


def test_state_dict_names_match_reference_default_model():
    """822-entry checkpoint compatibility starts with the IntVOS-owned names."""
    from cvpr2020_manet_amd.config import make_cfg
    from cvpr2020_manet_amd.networks import IntVOS as M

    class Stub(nn.Module):
        def forward(self, x):
            return x

    model = M.IntVOS(make_cfg(["--TEST_MODE", "True"]), Stub())
    g = load_golden("statedict_default")
    keys = sorted(model.state_dict().keys())
    assert keys == g["keys"].tolist()
    shapes = [str(tuple(model.state_dict()[k].shape)) for k in keys]
    assert shapes == g["shapes"].tolist()
    # aliased parameters really are the same tensors (IntVOS.py:537-543)
    assert model.semantic_embedding[0].weight is model.seperate_conv.weight
    assert model.semantic_embedding[3].weight is model.embedding_conv.weight


def _oracle_backed_ops(monkeypatch, oracle):
    from cvpr2020_manet_amd import ops

    def fake_global(reference_embeddings, query_embeddings, reference_labels, n_ids, k_nearest_neighbors=1,
                    compute="f32", normalize=False, mem=None):
        C = query_embeddings.shape[-1]
        raw = oracle.global_match(reference_embeddings.reshape(-1, 1, C).numpy(),
                                  query_embeddings.reshape(-1, 1, C).numpy(),
                                  reference_labels.reshape(-1, 1, 1).numpy(), k_nearest_neighbors, n_ids=n_ids)
        gq, m = oracle.normalize_merge(raw.reshape(-1, n_ids), None if mem is None else mem.numpy().reshape(-1, n_ids),
                                       normalize=normalize)
        if mem is not None:
            mem.copy_(torch.from_numpy(m).view_as(mem))
        return torch.from_numpy(gq)

    def fake_local(prev, cur, labels, n_ids, max_distance=12, downsample=True):
        h, w, _ = cur.shape
        return torch.from_numpy(oracle.local_match(prev.numpy(), cur.numpy(), labels.numpy(), n_ids, max_distance,
                                                   downsample).reshape(h, w, n_ids))

    def fake_merge(x, mem=None, normalize=True):
        gq, m = oracle.normalize_merge(x.numpy().reshape(-1), None if mem is None else mem.numpy().reshape(-1), normalize)
        x.copy_(torch.from_numpy(gq).view_as(x))
        if mem is not None:
            mem.copy_(torch.from_numpy(m).view_as(mem))
        return x

    monkeypatch.setattr(ops, "global_match", fake_global)
    monkeypatch.setattr(ops, "local_match", fake_local)
    monkeypatch.setattr(ops, "normalize_merge_", fake_merge)


def test_host_logic_on_cpu_with_oracle_backed_ops(monkeypatch, oracle):
    _oracle_backed_ops(monkeypatch, oracle)
    g = load_golden("e2e_tiny")
    model = build_model(g, "cpu")
    compare(run_script(model, g, "cpu"), g, tol_logits=2e-4)


def _session(model, embs, scribs, rounds, nobj, dist_buffer):
    """one interactive session on a sequence as test.py:121-124,313-314 runs it: FRESH dicts, `rounds` = annotated frame per
    round; the [104,9] weight table lives in `dist_buffer` (the caller's memory: a new tensor object at the same address)"""
    seq, F_ = "clip", embs.shape[0]
    dist_buffer[:] = 0
    gmap, lmaps = {}, ({}, {seq: torch.from_numpy(dist_buffer)})
    gt = torch.Tensor([nobj])
    H, W = scribs[0].shape[-2:]
    up = lambda x: torch.argmax(nn.functional.interpolate(x, size=(H, W), mode="bilinear", align_corners=True), dim=1)
    logits, prev_round = {}, None
    for r, start in enumerate(rounds, 1):
        tmp, lmaps = model.int_seghead(ref_frame_embedding=embs[start:start + 1], ref_scribble_label=scribs[r - 1],
                                       prev_round_label=prev_round, global_map_tmp_dic=gmap, local_map_dics=lmaps,
                                       interaction_num=r, seq_names=[seq], gt_ids=gt, frame_num=[start], first_inter=r == 1)
        prev_round = up(tmp[seq]).float().unsqueeze(0)
        for order in (range(start + 1, F_), range(start - 1, -1, -1)):
            prev_label, prev_emb = up(tmp[seq]).unsqueeze(0), embs[start:start + 1]
            for ii in order:
                out, gmap, lmaps = model.prop_seghead(embs[start:start + 1], prev_emb, embs[ii:ii + 1], scribs[r - 1], prev_label,
                                                      seq_names=[seq], gt_ids=gt, global_map_tmp_dic=gmap, local_map_dics=lmaps,
                                                      interaction_num=r, start_annotated_frame=start, frame_num=[ii],
                                                      dynamic_seghead=model.dynamic_seghead)
                logits[(r, ii)] = out[seq].clone()
                prev_label, prev_emb = up(out[seq]).unsqueeze(0), embs[ii:ii + 1]
    return logits, lmaps


def test_second_session_on_a_sequence_does_not_see_the_first_sessions_weights(monkeypatch, oracle):
    """the host mirror of local_map_dist_dic was keyed on the table's ADDRESS.  test.py drops the dicts after
    a session and builds fresh ones for the next session on the same sequence name; the new [104,9] table often lands on the
    freed address, the old session's weights then counted as this session's -- frame 2 in round 2 of sessions [0] -> [2, 0]
    got the placeholder slot instead of its new local map.  Here the second session's table IS at the first one's address
    (same caller-owned buffer, new tensor object): its results must equal those of a model that never saw the first session."""
    _oracle_backed_ops(monkeypatch, oracle)
    g = load_golden("e2e_tiny")
    t = lambda a: torch.from_numpy(a)
    scribs = [t(g["scrib"]), t(g["scrib2"])]
    nobj = int(g["nobj"])
    buf = np.zeros((104, 9), dtype=np.float32)
    with torch.no_grad():
        seen = build_model(g, "cpu")
        embs = seen.extract_feature(t(g["imgs"]))
        assert embs.shape[0] >= 4
        _session(seen, embs, scribs, [0], nobj, buf)
        got, lm = _session(seen, embs, scribs, [2, 0], nobj, buf)
        assert lm[1]["clip"].data_ptr() == buf.ctypes.data  # (the address-reuse case, deterministically)
        fresh = build_model(g, "cpu")
        want, _ = _session(fresh, embs, scribs, [2, 0], nobj, np.zeros((104, 9), dtype=np.float32))
    assert sorted(got) == sorted(want) and (2, 2) in got
    for k in want:
        assert torch.equal(got[k], want[k]), k
    # an external write to the table (not through the module) drops the mirror instead of trusting it
    tab = lm[1]["clip"]
    m = seen._mirror_of("clip", tab)
    assert m is not None and m[2]
    tab.zero_()
    assert seen._mirror_of("clip", tab)[2] == {}


def test_cpu_tensors_are_refused_by_the_product_path():
    from cvpr2020_manet_amd.networks import IntVOS as M
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        M.nearest_neighbor_features_per_object(torch.zeros(4, 4, 8), torch.zeros(4, 4, 8),
                                               torch.zeros(4, 4, 1, dtype=torch.int32), 1, gt_ids=torch.tensor(1.))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        M.local_previous_frame_nearest_neighbor_features_per_object(
            torch.zeros(4, 4, 8), torch.zeros(4, 4, 8), torch.zeros(4, 4, 1, dtype=torch.int32),
            torch.arange(2).int(), max_distance=1)


@pytest.mark.gpu
def test_end_to_end_on_gpu_matches_reference_class():
    g = load_golden("e2e_tiny")
    model = build_model(g, "cuda")
    # logits pass through MIOpen convolutions: fp32 rounding of a different conv algorithm
    compare(run_script(model, g, "cuda"), g, tol_logits=5e-4)


@pytest.mark.gpu
def test_end_to_end_on_gpu_prepares_every_frame_once():
    """the propagation script touches 4 frames (3 of them as query, 2 as annotated frame): each is prepared exactly once
    (one launch each), whatever the number of times it is the current / previous / annotated frame or the round"""
    from cvpr2020_manet_amd import ops
    g = load_golden("e2e_tiny")
    model = build_model(g, "cuda")
    calls = []
    real = ops.prepare_frames

    def counting(emb, **kw):
        calls.append(tuple(emb.shape))
        return real(emb, **kw)
    ops.prepare_frames = counting
    try:
        out = run_script(model, g, "cuda")
    finally:
        ops.prepare_frames = real
    compare(out, g, tol_logits=5e-4)
    # run_script: frames 1, 2, 3 of `embs` + the previous / current frame of the batch model.forward extracts itself
    assert len(calls) == 5 and all(len(c) == 3 for c in calls)
    # a driver that takes the producer's fused epilogue (extract_feature(packed=True) -> ops.embed_finish: bn2 + relu2 + cast +
    # every frame's operands in one launch per batch) leaves the loop with nothing to prepare
    model2 = build_model(g, "cuda")
    real_extract = model2.extract_feature
    model2.extract_feature = lambda x: real_extract(x, packed=True)
    calls.clear()
    fused_calls = []
    real_finish = ops.embed_finish

    def counting_finish(conv_out, *a, **kw):
        fused_calls.append(int(conv_out.shape[0]))
        return real_finish(conv_out, *a, **kw)
    ops.prepare_frames, ops.embed_finish = counting, counting_finish
    try:
        out2 = run_script(model2, g, "cuda")
    finally:
        ops.prepare_frames, ops.embed_finish = real, real_finish
    assert fused_calls == [4, 3] and calls == []  # two fused launches: the clip of 4, forward()'s own batch of 3
    compare(out2, g, tol_logits=5e-4)  # (BatchNorm folded into one fmaf: last-place differences in the embeddings)


@pytest.mark.gpu
def test_end_to_end_on_gpu_with_2_byte_embeddings_and_bf16_arithmetic():
    """SURVEY 8f rank 4 through the drop-in module: IntVOS(cfg, fe, compute=..., emb_dtype="bf16") stores
    extract_feature's output in bf16, the matching kernels read 2-byte embeddings (and, for compute="bf16", multiply in
    bf16), the heads widen them.  Bounded against the reference class's fp32 logits (e2e_tiny.npz): the embeddings carry
    8 significand bits, the logits move by a few 1e-3 of their range."""
    g = load_golden("e2e_tiny")
    ref_scale = max(float(np.abs(g[k]).max()) for k in ("int_logits", "prop1_logits_3", "prop2_logits_3"))
    for compute in ("f32", "bf16", "bf16x3"):
        model = build_model(g, "cuda", compute=compute, emb_dtype="bf16")
        out = run_script(model, g, "cuda")
        assert out["embs"].dtype == torch.bfloat16
        np.testing.assert_allclose(out["embs"].float().cpu().numpy(), g["embs"], rtol=2 ** -8, atol=1e-6)
        for k in ("gmap_round1", "gmap_round2", "lmap_tmp"):
            err = float(np.abs(out[k].cpu().numpy() - g[k]).max())
            assert err < 1e-2, (compute, k, err)  # normalised maps in [0, 1]
        for k in ("int_logits", "prop1_logits_2", "prop1_logits_3", "int2_logits", "prop2_logits_3"):
            err = float(np.abs(out[k].cpu().numpy() - g[k]).max())
            assert err < 2e-2 * max(ref_scale, 1.0), (compute, k, err)


@pytest.mark.gpu
def test_end_to_end_bf16r_equals_f32_through_the_module():
    """IntVOS(cfg, fe, compute="bf16r"): the bf16 filter + exact fp32 re-rank behind the reference API -- every map and every
    logit of the scripted session (two interaction rounds, propagation both ways) EQUALS the compute="f32" model's, bit for
    bit, and both sit on the reference class's fixture."""
    g = load_golden("e2e_tiny")
    outs = {}
    for compute in ("f32", "bf16r"):
        outs[compute] = run_script(build_model(g, "cuda", compute=compute), g, "cuda")
    for k in ("gmap_round1", "gmap_round2", "lmap_tmp", "int_logits", "prop1_logits_2", "prop1_logits_3", "int2_logits",
              "prop2_logits_3"):
        assert torch.equal(outs["bf16r"][k], outs["f32"][k]), k
        np.testing.assert_allclose(outs["bf16r"][k].cpu().numpy(), g[k], rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
def test_module_functions_on_gpu_match_reference():
    from cvpr2020_manet_amd.networks import IntVOS as M
    M.set_cfg(tiny_cfg())
    g = load_golden("global_k1_tm1")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out, ids = M.nearest_neighbor_features_per_object(dev(g["ref_chw"]).permute(1, 2, 0), dev(g["qry_chw"]).permute(1, 2, 0),
                                                      dev(g["labels"]), 1, gt_ids=torch.tensor(3.), n_chunks=7)
    assert out.shape == g["out"].shape and ids.dtype == torch.int32 and ids.tolist() == g["ids"].tolist()
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-5, atol=2e-6)
    g = load_golden("global_k1_noids_tm1")
    out, ids = M.nearest_neighbor_features_per_object(dev(g["ref_chw"]).permute(1, 2, 0), dev(g["qry_chw"]).permute(1, 2, 0),
                                                      dev(g["labels"]), 1, gt_ids=None)
    assert ids.tolist() == g["ids"].tolist()
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-5, atol=2e-6)
    g = load_golden("local_ds1_C16_12x15_d2")
    out = M.local_previous_frame_nearest_neighbor_features_per_object(
        dev(g["prev_chw"]).permute(1, 2, 0), dev(g["cur_chw"]).permute(1, 2, 0), dev(g["labels"]),
        torch.arange(int(g["n_ids"])).int().cuda(), max_distance=int(g["d"]))
    assert out.shape == g["out"].shape
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-5, atol=2e-6)
    dist = M.local_pairwise_distances2(dev(g["cur_chw"]).permute(1, 2, 0), dev(g["prev_chw"]).permute(1, 2, 0),
                                       max_distance=int(g["d"]))
    np.testing.assert_allclose(dist.cpu().numpy(), g["dist"], rtol=1e-5, atol=2e-6)


def test_cfg_defaults_equal_reference():
    """same flag names and defaults as the reference's config.py:17-80 (fixture from oracle/gen_golden.py)"""
    import json
    import os
    from conftest import GOLDEN
    from cvpr2020_manet_amd.config import make_cfg
    ref = json.load(open(os.path.join(GOLDEN, "config_defaults.json")))
    # (MODEL_MATCH_COMPUTE / MODEL_EMB_DTYPE / MODEL_HEAD_POINTWISE / MODEL_CACHE_FRAMES are this implementation's
    # extension flags, absent from the reference)
    ext = ("MODEL_MATCH_COMPUTE", "MODEL_EMB_DTYPE", "MODEL_HEAD_POINTWISE", "MODEL_CACHE_FRAMES", "MODEL_LOCAL_VOLUME_CACHE_MB",
           "MODEL_LOCAL_VOLUME_LAZY", "MODEL_HEAD_MEMO_MB")
    mine = {k: v for k, v in vars(make_cfg([])).items() if k != "ROOT_DIR" and k not in ext}
    assert mine == ref
    assert make_cfg([]).MODEL_MATCH_COMPUTE == "f32" and make_cfg([]).MODEL_EMB_DTYPE == "f32"
    assert make_cfg([]).MODEL_HEAD_POINTWISE == "f32" and make_cfg([]).MODEL_CACHE_FRAMES is True  # exact head by default
    assert make_cfg([]).MODEL_LOCAL_VOLUME_CACHE_MB == 8192 and make_cfg([]).MODEL_LOCAL_VOLUME_LAZY is False
    assert make_cfg([]).MODEL_HEAD_MEMO_MB == 8192
    assert make_cfg(["--TEST_MODE", "True", "--unknown-flag", "1"]).TEST_MODE is True


def test_constructor_switches_and_cache_hooks_on_cpu():
    """IntVOS(cfg, fe, compute=..., emb_dtype=...) / cfg.MODEL_MATCH_COMPUTE / cfg.MODEL_EMB_DTYPE, and
    the cache invalidation hooks: host logic only, no device work."""
    from cvpr2020_manet_amd.config import make_cfg
    from cvpr2020_manet_amd.networks import IntVOS as M
    m = M.IntVOS(tiny_cfg(), TinyExtractor())
    assert m.compute == "f32" and m.emb_dtype == torch.float32
    m = M.IntVOS(tiny_cfg(), TinyExtractor(), compute="bf16r", emb_dtype="bf16")
    assert m.compute == "bf16r" and m.emb_dtype == torch.bfloat16
    cfg = make_cfg(["--TEST_MODE", "True", "--MODEL_SEMANTIC_EMBEDDING_DIM", "12", "--MODEL_HEAD_EMBEDDING_DIM", "8",
                    "--MODEL_ASPP_OUTDIM", "6", "--MODEL_MATCH_COMPUTE", "bf16x3", "--MODEL_EMB_DTYPE", "bf16"])
    m = M.IntVOS(cfg, TinyExtractor())
    assert m.compute == "bf16x3" and m.emb_dtype == torch.bfloat16
    ref_cfg = argparse.Namespace(**{k: v for k, v in vars(tiny_cfg()).items()
                                    if k not in ("MODEL_MATCH_COMPUTE", "MODEL_EMB_DTYPE", "MODEL_HEAD_POINTWISE",
                                                 "MODEL_CACHE_FRAMES")})
    mref = M.IntVOS(ref_cfg, TinyExtractor())  # a reference cfg object without the extension flags
    assert mref.compute == "f32" and mref.pointwise == "f32" and mref.cache_frames is True
    # the heads' 1x1 arithmetic is per model: two models in one process may differ, exact fp32 by default
    ma, mb = M.IntVOS(tiny_cfg(), TinyExtractor()), M.IntVOS(tiny_cfg(), TinyExtractor(), pointwise="split")
    assert M._pointwise_mode(ma.dynamic_seghead.layer2) == "f32" and M._pointwise_mode(mb.dynamic_seghead.layer2) == "split"
    assert M._pointwise_mode(M._split_separable_conv2d(4, 4)) == "f32"  # a stand-alone block: the module default
    assert M.IntVOS(tiny_cfg(), TinyExtractor(), cache_frames=False).cache_frames is False
    with pytest.raises(ValueError):
        M.IntVOS(tiny_cfg(), TinyExtractor(), pointwise="fp8")
    with pytest.raises(ValueError):
        M.IntVOS(tiny_cfg(), TinyExtractor(), compute="fp8")
    with pytest.raises(ValueError):
        M.IntVOS(tiny_cfg(), TinyExtractor(), emb_dtype="f16")
    # extract_feature stores in emb_dtype (inference), keeps fp32 under autograd
    with torch.no_grad():
        assert m.extract_feature(torch.zeros(1, 3, 16, 16)).dtype == torch.bfloat16
    assert m.extract_feature(torch.zeros(1, 3, 16, 16)).dtype == torch.float32
    # the identity-keyed caches are dropped by train(), load_state_dict(), _apply() and invalidate_caches()
    for hook in (lambda: m.train(), lambda: m.eval(), lambda: m.load_state_dict(m.state_dict()), lambda: m.float(),
                 m.invalidate_caches):
        m._bank_cache["s"] = ("key", None)
        m._frame_cache["k"] = object()
        m.dynamic_seghead.layer1._fold_cache = ("stale", {})
        m._vol_cache[("prev", "cur")] = [torch.zeros(1), None, None]  # (r6: the stored local-match volumes go with them)
        m._vol_cache_bytes, m._memo_bytes = 4, 104
        hook()
        assert not m._bank_cache and not m._frame_cache and m.dynamic_seghead.layer1._fold_cache is None
        assert not m._vol_cache and m.local_volume_bytes_cached() == 0 and m._memo_bytes == 0
