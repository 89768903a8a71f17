"""the path takes the producer's storage -- bf16 embeddings read as 2-byte words end to end,
query operand images packed once per frame (PackedQuery), the annotated frame's bank prepared once per interaction
by the drop-in module.  All of it must be bit-identical to the plain fp32-tensor calls on the same VALUES."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cvpr2020_manet_amd import ops as o
    return o


def _emb(seed, C, h, w):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.relu(torch.randn(C, h, w, generator=g, device="cuda")) * 0.1


@pytest.mark.parametrize("compute", ["bf16", "bf16x3", "f32"])
def test_bf16_stored_embeddings_equal_widened_fp32(ops, compute):
    q, k = _emb(1, 100, 30, 41), _emb(2, 100, 60, 41)
    lab = torch.randint(-1, 4, (60 * 41,), device="cuda", dtype=torch.int32)
    qb, kb = q.bfloat16(), k.bfloat16()
    got = ops.global_match(kb.permute(1, 2, 0), qb.permute(1, 2, 0), lab, 4, compute=compute)
    want = ops.global_match(kb.float().permute(1, 2, 0), qb.float().permute(1, 2, 0), lab, 4, compute=compute)
    assert torch.equal(got, want)
    # mixed storage is fine too (bank bf16, query fp32)
    mixed = ops.global_match(kb.permute(1, 2, 0), qb.float().permute(1, 2, 0), lab, 4, compute=compute)
    assert torch.equal(mixed, want)


@pytest.mark.parametrize("compute", ["bf16", "f32"])
def test_packed_query_and_prepared_bank(ops, compute):
    q, k = _emb(3, 100, 30, 53), _emb(4, 100, 90, 53)
    lab = torch.randint(0, 3, (90 * 53,), device="cuda", dtype=torch.int32)
    src = (lambda t: t.bfloat16()) if compute == "bf16" else (lambda t: t)
    bank = ops.PreparedBank(src(k).permute(1, 2, 0), lab, 3, compute=compute)
    plain = bank.match(src(q).permute(1, 2, 0), normalize=True)
    pq = ops.PackedQuery(src(q).permute(1, 2, 0), compute=compute)
    mem = torch.ones(30 * 53, 3, device="cuda")
    packed = bank.match(pq, normalize=True, mem=mem)
    assert torch.equal(plain, packed) and torch.equal(mem, packed)
    one_shot = ops.global_match(src(k).permute(1, 2, 0), src(q).permute(1, 2, 0), lab, 3, compute=compute, normalize=True)
    assert torch.equal(plain, one_shot)
    with pytest.raises(ValueError, match="was packed for"):
        ops.PreparedBank(k.permute(1, 2, 0), lab, 3, compute="bf16x3").match(pq)


@pytest.mark.parametrize("d", [2, 4, 12])
def test_local_match_reads_bf16_embeddings(ops, oracle, d):
    prev, cur = _emb(5, 100, 31, 45).bfloat16(), _emb(6, 100, 31, 45).bfloat16()
    lab = torch.randint(0, 3, (31, 45), device="cuda", dtype=torch.int32)
    got = ops.local_match(prev.permute(1, 2, 0), cur.permute(1, 2, 0), lab, 3, d)
    want = ops.local_match(prev.float().permute(1, 2, 0), cur.float().permute(1, 2, 0), lab, 3, d)
    assert torch.equal(got, want)
    ref = oracle.local_match(prev.float().permute(1, 2, 0).cpu().numpy(), cur.float().permute(1, 2, 0).cpu().numpy(),
                             lab.cpu().numpy(), 3, d).reshape(31, 45, 3)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=2e-6)


def test_module_prepares_the_bank_once_per_interaction(ops):
    """prop_seghead called frame after frame with the same annotated frame (test.py:237-259): one PreparedBank;
    a new scribble (in-place edit or another tensor) -> a new bank; results identical to the one-shot op."""
    import torch.nn as nn
    from cvpr2020_manet_amd.config import make_cfg
    from cvpr2020_manet_amd.networks import IntVOS as M

    class Stub(nn.Module):
        def forward(self, x):
            return x
    cfg = make_cfg(["--TEST_MODE", "True", "--MODEL_SEMANTIC_EMBEDDING_DIM", "16", "--MODEL_HEAD_EMBEDDING_DIM", "8",
                    "--MODEL_ASPP_OUTDIM", "6", "--MODEL_MAX_LOCAL_DISTANCE", "2"])
    model = M.IntVOS(cfg, Stub()).cuda().eval()
    embs = torch.stack([_emb(10 + i, 16, 12, 14) for i in range(4)])
    scrib = torch.full((1, 1, 12, 14), -1.0, device="cuda")
    scrib[0, 0, 2:5, 3:9] = 1
    scrib[0, 0, 8:10, :] = 0
    prev_label = torch.zeros(1, 1, 48, 56, device="cuda")
    gmap = {}
    seen = []
    with torch.no_grad():
        for ii in (1, 2, 3):
            dic, gmap = model.prop_seghead(embs[0:1], embs[ii - 1:ii], embs[ii:ii + 1], scrib, prev_label, True, True,
                                           ["s"], torch.Tensor([1]), 1, gmap, None, 1, 0, [ii], model.dynamic_seghead)
            seen.append(model._bank_cache["s"][1])
            want = ops.global_match(embs[0].permute(1, 2, 0), embs[ii].permute(1, 2, 0), scrib[0].permute(1, 2, 0).int(), 2,
                                    normalize=True)
            assert torch.equal(gmap["s"][ii].reshape(-1, 2), want)  # first use of the slot: merged with ones == itself
        assert seen[0] is seen[1] is seen[2]
        scrib[0, 0, 0, 0] = 1  # in-place edit bumps the version counter
        model.prop_seghead(embs[0:1], embs[0:1], embs[1:2], scrib, prev_label, True, True, ["s"], torch.Tensor([1]), 1, gmap,
                           None, 1, 0, [1], model.dynamic_seghead)
        assert model._bank_cache["s"][1] is not seen[0]


@pytest.mark.parametrize("storage", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("compute", ["f32", "bf16", "bf16r", "bf16x3"])
@pytest.mark.parametrize("shape,d", [((3, 100, 120, 214), 12), ((2, 100, 21, 30), 4), ((1, 100, 12, 15), 2), ((2, 37, 9, 16), -1)])
def test_embed_finish_is_bn_relu_cast_plus_frame_prepare_in_one_launch(ops, storage, compute, shape, d):
    """manet_embed_finish: eval-mode bn2 + relu2 + the storage cast + manet_frame_prepare behind the
    framework's 1x1 embedding GEMM (IntVOS.py:537-543).  The embedding it stores equals the stock elementwise chain (BatchNorm
    folded into one fmaf: a rounding of the last place); the frame operands are BIT-IDENTICAL to manet_frame_prepare run on
    that stored embedding (so everything downstream -- global match, local match -- is the separate route's bits)."""
    B, C, h, w = shape
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + h)
    conv_out = torch.randn(B, C, h, w, generator=g, device="cuda") * 0.3
    bn = torch.nn.BatchNorm2d(C).cuda().eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.2, generator=g); bn.running_var.uniform_(0.5, 2.0, generator=g)
        bn.weight.normal_(1, 0.2, generator=g); bn.bias.normal_(0, 0.2, generator=g)
        want = torch.relu(bn(conv_out)).to(storage)
        scale, shift = ops.fold_bn(bn)
        emb, frames = ops.embed_finish(conv_out, scale, shift, relu=True, emb_dtype=storage, compute=compute, max_distance=d)
    assert emb.dtype == storage and emb.shape == conv_out.shape and len(frames) == B
    # folded BatchNorm: one fmaf instead of (x - mean) * invstd * weight + bias -- last-place differences, one bf16 ulp after rounding
    tol = dict(rtol=1e-5, atol=1e-6) if storage == torch.float32 else dict(rtol=2 ** -7, atol=1e-6)
    torch.testing.assert_close(emb.float(), want.float(), **tol)
    ref_frames = ops.prepare_frames(emb, compute=compute, max_distance=d)
    _same_operands(ops, frames, ref_frames, emb, compute, d)
    # a strided batch view of the conv output (channels-last-like storage) goes through the generic-stride path
    conv_cl = conv_out.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    emb2, frames2 = ops.embed_finish(conv_cl, scale, shift, relu=True, emb_dtype=storage, compute=compute, max_distance=d)
    assert torch.equal(emb2, emb)
    _same_operands(ops, frames2, ref_frames, emb, compute, d)


def _same_operands(ops, frames, ref_frames, emb, compute, d):
    """two sets of prepared frames carry the same operands: the global match against one bank and the local match between
    neighbouring frames give the same bits (the workspaces' alignment gaps are never written, so bytes cannot be compared)"""
    B, C, h, w = emb.shape
    g = torch.Generator(device="cuda").manual_seed(99)
    n_ids = 3
    bank_rows = (torch.relu(torch.randn(700, C, generator=g, device="cuda")) * 0.3).to(emb.dtype)
    bank_lab = torch.randint(0, n_ids, (700,), generator=g, device="cuda", dtype=torch.int32)
    bank = ops.PreparedBank(bank_rows, bank_lab, n_ids, compute=compute)
    lab = torch.randint(0, n_ids, (h, w), generator=g, device="cuda", dtype=torch.int32)
    for i, (a, b) in enumerate(zip(frames, ref_frames)):
        assert torch.equal(bank.match(a), bank.match(b)), "query operand image"
        if d >= 0:
            j = (i + 1) % B
            assert torch.equal(ops.local_match_frames(frames[j], a, lab, n_ids), ops.local_match_frames(ref_frames[j], b, lab, n_ids)), "pooled plane"


def test_extract_feature_packed_route_feeds_the_frame_cache(ops):
    """IntVOS.extract_feature(packed=True) in inference: embeddings equal the stock route's (to BatchNorm-folding rounding), every
    frame's operands are in the model's cache afterwards (no manet_frame_prepare launch in the propagation loop)."""
    from cvpr2020_manet_amd.config import make_cfg
    from cvpr2020_manet_amd.networks.IntVOS import IntVOS

    class Enc(torch.nn.Module):
        def __init__(self, out_dim):
            super().__init__()
            self.net = torch.nn.Conv2d(3, out_dim, 3, stride=4, padding=1)

        def forward(self, x):
            return self.net(x)

    torch.manual_seed(5)
    cfg = make_cfg(["--TEST_MODE", "True"])
    for emb_dtype in ("f32", "bf16"):
        model = IntVOS(cfg, Enc(cfg.MODEL_ASPP_OUTDIM), compute="bf16" if emb_dtype == "bf16" else "f32", emb_dtype=emb_dtype).cuda().eval()
        with torch.no_grad():
            for m in (model.bn1, model.bn2):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 2.0)
            x = torch.randn(3, 3, 96, 128, device="cuda")
            stock = model.extract_feature(x)
            assert len(model._frame_cache) == 0
            fused = model.extract_feature(x, packed=True)
            assert fused.dtype == stock.dtype
            tol = dict(rtol=1e-4, atol=1e-5) if emb_dtype == "f32" else dict(rtol=2 ** -7, atol=1e-5)
            torch.testing.assert_close(fused.float(), stock.float(), **tol)
            assert len(model._frame_cache) == 3
            hit, _ = model._prepared_frame(fused[1])
            assert len(model._frame_cache) == 3  # (a hit: nothing new was prepared)
            fresh = ops.prepare_frames(fused, compute=model.compute, max_distance=model._local_radius())
            _same_operands(ops, [model._prepared_frame(fused[i])[0] for i in range(3)], fresh, fused, model.compute, model._local_radius())


@pytest.mark.gpu
def test_embedding_head_depthwise_stage_in_one_launch(ops):
    """r5: relu1(bn1(seperate_conv(x))) (IntVOS.py:537-539) through the head's depthwise kernel (the 3x3 taps in a zero-padded 7x7):
    equals the module sequence to summation-order rounding, follows an in-place weight update, and leaves training mode / other
    layer shapes to the modules"""
    from cvpr2020_manet_amd.config import make_cfg
    from cvpr2020_manet_amd.networks.IntVOS import IntVOS
    torch.manual_seed(9)
    cfg = make_cfg(["--TEST_MODE", "True"])
    model = IntVOS(cfg, torch.nn.Identity()).cuda().eval()
    with torch.no_grad():
        model.bn1.running_mean.normal_(0, 0.2); model.bn1.running_var.uniform_(0.5, 2.0)
        model.bn1.weight.normal_(1, 0.2); model.bn1.bias.normal_(0, 0.2)
        for shape in ((2, cfg.MODEL_ASPP_OUTDIM, 30, 54), (1, cfg.MODEL_ASPP_OUTDIM, 7, 9)):
            x = torch.randn(*shape, device="cuda")
            want = model.relu1(model.bn1(model.seperate_conv(x)))
            calls = {"n": 0}
            real = ops.dwconv7x7_bn_relu

            def counting(*a, **k):
                calls["n"] += 1
                return real(*a, **k)
            ops.dwconv7x7_bn_relu = counting
            try:
                got = model._separate_conv_bn_relu(x)
            finally:
                ops.dwconv7x7_bn_relu = real
            assert calls["n"] == 1
            torch.testing.assert_close(got, want, rtol=1e-5, atol=2e-6)
        model.seperate_conv.weight.mul_(0.5)  # in place: the padded copy is rebuilt
        torch.testing.assert_close(model._separate_conv_bn_relu(x), model.relu1(model.bn1(model.seperate_conv(x))), rtol=1e-5, atol=2e-6)
        model.train()
        calls = {"n": 0}
        ops.dwconv7x7_bn_relu = counting
        try:
            model._separate_conv_bn_relu(x)
        finally:
            ops.dwconv7x7_bn_relu = real
        assert calls["n"] == 0  # training mode: the modules (batch statistics)
