"""the fused depthwise-7x7 + BN + ReLU kernel and the head built on it, against the
reference's own _split_separable_conv2d / DynamicSegHead (golden seg_head_tiny.npz, eval mode)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

# conv tap order differs between the reference's conv backend and the kernel/oracle: fp32 rounding
RTOL, ATOL = 1e-5, 1e-5


def folded_bn(g, prefix):
    inv = 1.0 / np.sqrt(g[prefix + "running_var"] + float(g["eps"]))
    scale = g[prefix + "weight"] * inv
    shift = g[prefix + "bias"] - g[prefix + "running_mean"] * scale
    return scale.astype(np.float32), shift.astype(np.float32)


def test_oracle_matches_reference_block(oracle):
    g = load_golden("seg_head_tiny")
    scale, shift = folded_bn(g, "blk::bn1.")
    half = oracle.dwconv7x7_bn_relu(g["x"], g["blk::conv1.weight"], g["blk::conv1.bias"], scale, shift, relu=True)
    np.testing.assert_allclose(half, g["half"], rtol=RTOL, atol=ATOL)


def make_block(g, device):
    from cvpr2020_manet_amd.config import make_cfg
    from cvpr2020_manet_amd.networks import IntVOS as M
    M.set_cfg(make_cfg(["--MODEL_SEMANTIC_EMBEDDING_DIM", "13", "--MODEL_HEAD_EMBEDDING_DIM", "24"]))
    blk = M._split_separable_conv2d(6, 10)
    blk.load_state_dict({k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("blk::")})
    head = M.DynamicSegHead()
    head.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("head::")})
    return blk.to(device).eval(), head.to(device).eval()


def test_module_on_cpu_is_the_stock_path():
    g = load_golden("seg_head_tiny")
    blk, head = make_block(g, "cpu")
    with torch.no_grad():
        np.testing.assert_allclose(blk(torch.from_numpy(g["x"])).numpy(), g["full"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(head(torch.from_numpy(g["hx"])).numpy(), g["hout"], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_gpu_kernel_vs_reference_and_oracle(oracle):
    from cvpr2020_manet_amd import ops
    g = load_golden("seg_head_tiny")
    blk, head = make_block(g, "cuda")
    x = torch.from_numpy(g["x"]).cuda()
    with pytest.raises(RuntimeError, match="requires grad"):  # Parameters + grad mode: would drop their gradient
        ops.dwconv7x7_bn_relu(x, blk.conv1.weight, blk.conv1.bias, blk.bn1)
    with torch.no_grad():
        half = ops.dwconv7x7_bn_relu(x, blk.conv1.weight, blk.conv1.bias, blk.bn1).cpu().numpy()
    np.testing.assert_allclose(half, g["half"], rtol=RTOL, atol=ATOL)
    scale, shift = folded_bn(g, "blk::bn1.")
    want = oracle.dwconv7x7_bn_relu(g["x"], g["blk::conv1.weight"], g["blk::conv1.bias"], scale, shift)
    np.testing.assert_allclose(half, want, rtol=1e-6, atol=1e-6)  # same tap order; scale/shift folded on device
    with torch.no_grad():  # the modules route through the kernel in eval mode on the GPU
        np.testing.assert_allclose(blk(x).cpu().numpy(), g["full"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(head(torch.from_numpy(g["hx"]).cuda()).cpu().numpy(), g["hout"], rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
def test_gpu_full_size_vs_framework_conv():
    """[3,256,120,214] (3 ids at 480p), ragged tile edges; against torch's own conv + BN + ReLU."""
    from cvpr2020_manet_amd import ops
    torch.manual_seed(1)
    C = 256
    conv = torch.nn.Conv2d(C, C, 7, padding=3, groups=C).cuda()
    bn = torch.nn.BatchNorm2d(C).cuda().eval()
    bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5)
    x = torch.randn(3, C, 120, 214, device="cuda")
    with torch.no_grad():
        want = torch.relu(bn(conv(x)))
        got = ops.dwconv7x7_bn_relu(x, conv.weight, conv.bias, bn)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 3, 61, 65), (3, 5, 9, 13), (2, 4, 130, 71), (4, 2, 7, 7), (2, 3, 60, 64), (3, 2, 121, 130),
                                   (2, 520, 9, 13), (3, 130, 61, 66), (3, 3, 120, 214)])
def test_depthwise_batch_walk_odd_widths_and_tile_edges(shape):
    """the depthwise kernel walks the batch items of a (tile, channel) with the next item's loads in flight (>= 512 workgroups:
    the two wide shapes) or takes a workgroup per (tile, channel, item) (fewer: the others, layer 1's 3 per-object channels):
    every item, odd and even widths (the 4-byte and the 8-byte load path), one / several tiles per axis, against torch's
    conv + BN (+ReLU)"""
    from cvpr2020_manet_amd import ops
    B, C, h, w = shape
    torch.manual_seed(B * 1000 + w)
    conv = torch.nn.Conv2d(C, C, 7, padding=3, groups=C).cuda()
    bn = torch.nn.BatchNorm2d(C).cuda().eval()
    bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5)
    x = torch.randn(B, C, h, w, device="cuda")
    with torch.no_grad():
        torch.testing.assert_close(ops.dwconv7x7_bn_relu(x, conv.weight, conv.bias, bn), torch.relu(bn(conv(x))), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(ops.dwconv7x7_bn_relu(x, conv.weight, conv.bias, bn, relu=False, relu_in=True),
                                   bn(conv(torch.relu(x))), rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_output_layer_and_deferred_relu_on_gpu():
    """ops.relu_conv1x1_c1 == Conv2d(C, 1, 1)(relu(x)); dwconv's relu_in == dwconv(relu(x)) bit for bit; and the whole
    DynamicSegHead fast path (ReLUs deferred into the next block, fused output layer) == the module's literal form."""
    import torch
    from cvpr2020_manet_amd import ops
    from cvpr2020_manet_amd.networks import IntVOS as M
    torch.manual_seed(11)
    for (B, C, h, w) in ((3, 256, 30, 54), (2, 7, 9, 13), (1, 5, 11, 10)):
        x = torch.randn(B, C, h, w, device="cuda")
        conv = torch.nn.Conv2d(C, 1, 1).cuda()
        with torch.no_grad():
            got = ops.relu_conv1x1_c1(x, conv.weight, conv.bias)
            want = conv(torch.relu(x))
            assert got.shape == want.shape
            torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(ops.relu_conv1x1_c1(x, conv.weight, None, relu_in=False),
                                       torch.nn.functional.conv2d(x, conv.weight), rtol=1e-5, atol=1e-5)
            dw = torch.nn.Conv2d(C, C, 7, padding=3, groups=C).cuda()
            a = ops.dwconv7x7_bn_relu(x, dw.weight, dw.bias, relu_in=True)
            b = ops.dwconv7x7_bn_relu(torch.relu(x), dw.weight, dw.bias)
            assert torch.equal(a, b)
    head = M.DynamicSegHead(in_dim=19, embed_dim=32).cuda().eval()
    for m in head.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(); m.running_var.uniform_(0.5, 2.0); m.weight.data.normal_(); m.bias.data.normal_()
    x = torch.randn(2, 19, 21, 34, device="cuda")
    with torch.no_grad():
        fast = head(x)
        ref = x
        for layer in (head.layer1, head.layer2, head.layer3, head.layer4):  # the reference's literal chain
            ref = layer.relu2(layer.bn2(layer.conv2(layer.relu1(layer.bn1(layer.conv1(ref))))))
        ref = head.conv(ref)
        torch.testing.assert_close(fast, ref, rtol=1e-3, atol=1e-3)
        shared = head.forward_shared(x[:1, :16], x[:, 16:])
        lit = head(torch.cat((x[:1, :16].repeat(2, 1, 1, 1), x[:, 16:]), 1))
        torch.testing.assert_close(shared, lit, rtol=1e-4, atol=1e-4)
    # r5: the blocks' ReLUs moved from the NEXT block's depthwise read (r2-r4) into their own 1x1 epilogue: the same bits
    head = M.DynamicSegHead(in_dim=103, embed_dim=256).cuda().eval()
    x = torch.randn(3, 103, 24, 36, device="cuda")
    with torch.no_grad():
        now = head(x)
        y = head.layer1(x, defer_relu=True)
        y = head.layer2(y, relu_in=True, defer_relu=True)
        y = head.layer3(y, relu_in=True, defer_relu=True)
        was = head.layer4(y, relu_in=True, defer_relu=True, head=(head.conv.weight, head.conv.bias))
        assert torch.equal(now, was)


@pytest.mark.gpu
def test_folded_constants_follow_parameter_updates():
    """The per-block cache of folded BatchNorm constants (r2) must notice in-place parameter / statistic updates,
    load_state_dict and device moves: same result as an uncached module every time."""
    import copy
    import torch
    from cvpr2020_manet_amd.networks import IntVOS as M
    torch.manual_seed(5)
    blk = M._split_separable_conv2d(12, 20).cuda().eval()
    x = torch.randn(2, 12, 17, 22, device="cuda")

    def literal(b):
        return b.relu2(b.bn2(b.conv2(b.relu1(b.bn1(b.conv1(x))))))

    with torch.no_grad():
        for step in range(4):
            if step == 1:
                blk.bn1.running_var.mul_(1.7); blk.bn2.weight.add_(0.3)          # in place
            if step == 2:
                sd = copy.deepcopy(blk.state_dict())
                for k in sd:
                    if sd[k].dtype.is_floating_point:
                        sd[k] = sd[k] * 0.9 + 0.05
                blk.load_state_dict(sd)                                          # copy_ into the same storage
            if step == 3:
                blk.conv2.weight.data = blk.conv2.weight.data.clone() * 1.1      # new storage
            torch.testing.assert_close(blk(x), literal(blk), rtol=1e-4, atol=1e-4)
            torch.testing.assert_close(blk(x), literal(blk), rtol=1e-4, atol=1e-4)  # second call: served from the cache


@pytest.mark.gpu
def test_split_bf16_pointwise_error_bound_and_forms():
    """ops.conv1x1_split (each fp32 factor = hi + lo bf16 pieces; hi*hi + hi*lo + lo*hi on the bf16 MFMA, fp32 accumulation)
    against an fp64 convolution.  Stated bound, per output:  |err| <= 2^-15 * sum_k |x_k| |w_k|  +  fp32 accumulation
    (2^-16 relative per product is the arithmetic's own bound: the dropped lo*lo term and the rounding of the lo pieces;
    measured on these cases: <= 0.2 of the bound, 2.5e-5 absolute on outputs of magnitude 7 at K = 256).  Any Cin (the
    per-object half of layer 1 has 3), partial pixel tiles, relu_out, the broadcast `add` term, the fused output layer."""
    import torch
    from cvpr2020_manet_amd import ops
    torch.manual_seed(5)
    with torch.no_grad():
        for (B, cin, h, w) in ((3, 256, 30, 54), (2, 100, 9, 12), (3, 3, 21, 36), (1, 1, 2, 2), (2, 17, 6, 10), (2, 33, 5, 52),
                               (3, 256, 120, 214)):
            x = torch.randn(B, cin, h, w, device="cuda") * 1.5
            w2t = torch.randn(cin, 256, device="cuda") * 0.08
            b2 = torch.randn(256, device="cuda")
            sw = ops.SplitWeight(w2t)
            w64 = w2t.t().reshape(256, cin, 1, 1).double()
            ref = torch.nn.functional.conv2d(x.double(), w64, b2.double())
            mag = torch.nn.functional.conv2d(x.double().abs(), w64.abs())
            got = ops.conv1x1_split(x, sw, b2)
            err = (got.double() - ref).abs()
            bound = 2.0 ** -15 * mag + 2.0 ** -22 * (mag + b2.abs().double().view(1, -1, 1, 1))
            assert bool((err <= bound).all()), (cin, float((err / bound).max()))
            assert torch.equal(ops.conv1x1_split(x, sw, b2, relu_out=True), torch.relu(got))
            add = torch.randn(1, 256, h, w, device="cuda")
            torch.testing.assert_close(ops.conv1x1_split(x, sw, b2, add=add), got + add, rtol=1e-6, atol=1e-6)
            torch.testing.assert_close(ops.conv1x1_split(x, sw, b2, add=add, relu_out=True), torch.relu(got + add), rtol=1e-6, atol=1e-6)
            fin = torch.nn.Conv2d(256, 1, 1).cuda()
            torch.testing.assert_close(ops.conv1x1_split(x, sw, b2, head_weight=fin.weight, head_bias=fin.bias),
                                       fin(torch.relu(got)), rtol=1e-4, atol=1e-4)
            if cin % 4 == 0:  # and against the exact fp32-MFMA kernel
                torch.testing.assert_close(got, ops.conv1x1_mfma(x, w2t, b2), rtol=1e-4, atol=1e-4)
        # non-finite inputs propagate as in fp32 (inf stays inf, NaN stays NaN), nothing else is touched
        x = torch.randn(1, 8, 4, 8, device="cuda")
        x[0, 3, 1, 2] = float("inf")
        x[0, 5, 2, 7] = float("nan")
        sw = ops.SplitWeight(torch.ones(8, 256, device="cuda"))
        y = ops.conv1x1_split(x, sw, torch.zeros(256, device="cuda"))
        bad = ~torch.isfinite(y[0, 0])
        assert bad.sum().item() == 2 and bool(bad[1, 2]) and bool(bad[2, 7])
        with pytest.raises(ValueError):
            ops.conv1x1_split(torch.randn(1, 8, 3, 3, device="cuda"), sw, torch.zeros(256, device="cuda"))  # h*w % 4
        with pytest.raises(TypeError):
            ops.conv1x1_split(x, torch.ones(8, 256, device="cuda"), torch.zeros(256, device="cuda"))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["split", "f32"])
def test_mfma_pointwise_vs_framework_conv_and_literal_module(mode):
    """ops.conv1x1_mfma (fp32-MFMA contraction fed by LDS-DMA) against the framework's 1x1 convolution with the same folded
    weights, and the blocks that use it against their literal form relu2(bn2(conv2(relu1(bn1(conv1(x)))))): full chunks,
    partial last chunks (Cin % 16 = 4, 8, 12; r4: any Cin), a partial last pixel tile, the full [3,256,120,214] size, the shared-embedding
    route of layer 1 (K = 100 on one batch item)."""
    import torch
    from cvpr2020_manet_amd import ops
    from cvpr2020_manet_amd.networks import IntVOS as M
    torch.manual_seed(3)
    keep = M.MFMA_POINTWISE
    M.MFMA_POINTWISE = mode  # the blocks below run their 1x1 stage on this kernel
    try:
        _pointwise_module_cases(M, ops, torch)
    finally:
        M.MFMA_POINTWISE = keep


def _pointwise_module_cases(M, ops, torch):
    with torch.no_grad():
        for (B, cin, h, w) in ((3, 256, 30, 54), (2, 100, 9, 12), (1, 16, 4, 16), (2, 4, 21, 36), (2, 44, 6, 10),
                               (3, 256, 120, 214)):
            blk = M._split_separable_conv2d(cin, 256).cuda().eval()
            for m in (blk.bn1, blk.bn2):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 2.0); m.weight.data.normal_(1, 0.2); m.bias.data.normal_(0, 0.2)
            x = torch.randn(B, cin, h, w, device="cuda")
            w2t, b2 = ops.fold_pointwise(blk.conv2, blk.bn2)
            want = torch.nn.functional.conv2d(x, w2t.t().reshape(256, cin, 1, 1).contiguous(), b2)
            got = ops.conv1x1_mfma(x, w2t, b2)
            torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)
            assert torch.equal(ops.conv1x1_mfma(x, w2t, b2, relu_out=True), torch.relu(got))
            lit = blk.relu2(blk.bn2(blk.conv2(blk.relu1(blk.bn1(blk.conv1(x))))))
            torch.testing.assert_close(blk(x), lit, rtol=2e-4, atol=2e-4)  # eval mode on the GPU: dw kernel + MFMA 1x1
        # fp64 spot check of the chain on the large case
        ref64 = torch.nn.functional.conv2d(x.double(), w2t.t().reshape(256, cin, 1, 1).double(), b2.double())
        assert float((got.double() - ref64).abs().max()) < 2e-4
        # layer 1's shared route: K = 100 over one batch item, then K = 3 per object with that added in the epilogue
        head = M.DynamicSegHead(in_dim=103, embed_dim=256).cuda().eval()
        xs = torch.randn(3, 103, 24, 30, device="cuda")
        shared = head.forward_shared(xs[:1, :100].contiguous(), xs[:, 100:].contiguous())
        lit = head(torch.cat((xs[:1, :100].repeat(3, 1, 1, 1), xs[:, 100:]), 1))
        torch.testing.assert_close(shared, lit, rtol=1e-3, atol=1e-3)
        # r4: any Cin (rows past Cin inside a 4-row DMA piece re-read row Cin - 1 against zero weight rows), and the `add` term
        for (B, cin, h, w) in ((3, 3, 24, 30), (1, 1, 4, 4), (2, 6, 5, 12), (2, 18, 9, 12), (2, 31, 9, 12), (3, 103, 6, 10),
                               (3, 3, 120, 214)):
            # (LDS is not cleared between kernels: leave NaNs in every stage row first -- a partial last chunk must not multiply
            # rows it never wrote by its zero weights; r4 did for Cin in 17..32)
            ops.conv1x1_mfma(torch.full((1, 48, 16, 64), float("nan"), device="cuda"), torch.zeros(48, 256, device="cuda"),
                             torch.zeros(256, device="cuda"))
            x = torch.randn(B, cin, h, w, device="cuda")
            w2t = torch.randn(cin, 256, device="cuda") * 0.2
            bb = torch.randn(256, device="cuda")
            addt = torch.randn(256, h, w, device="cuda")
            want = torch.nn.functional.conv2d(x, w2t.t().reshape(256, cin, 1, 1).contiguous(), bb)
            torch.testing.assert_close(ops.conv1x1_mfma(x, w2t, bb), want, rtol=2e-4, atol=2e-4)
            got = ops.conv1x1_mfma(x, w2t, bb, relu_out=True, add=addt)
            torch.testing.assert_close(got, (want + addt).relu(), rtol=2e-4, atol=2e-4)
            assert torch.equal(ops.conv1x1_mfma(x, w2t, bb, add=addt.unsqueeze(0)).relu(), got)
        # an infinite activation in the LAST channel stays what the contraction makes of it (+-inf by the weight's sign, no NaN
        # from a zero weight row meeting a copy of it): the rows past Cin are not fetched
        xi = torch.randn(2, 3, 8, 12, device="cuda")
        xi[1, 2, 3, 5] = float("inf")
        wi = torch.randn(3, 256, device="cuda").abs() + 0.1
        yi = ops.conv1x1_mfma(xi, wi, torch.zeros(256, device="cuda"))
        assert bool(torch.isinf(yi[1, :, 3, 5]).all()) and bool((yi[1, :, 3, 5] > 0).all()) and not bool(torch.isnan(yi).any())
        with pytest.raises(ValueError):
            ops.conv1x1_mfma(x, w2t, bb, add=addt[:, :4])
        with pytest.raises(ValueError):
            ops.conv1x1_mfma(x, w2t, bb, add=addt, head_weight=torch.zeros(1, 256, 1, 1, device="cuda"))
        with pytest.raises(ValueError):
            ops.conv1x1_mfma(torch.randn(1, 6, 3, 3, device="cuda"), torch.zeros(6, 256, device="cuda"), b2)  # h*w % 4
        # the head's output layer fused into layer4's pointwise kernel: Conv2d(256, 1, 1)(relu(z)) without z in memory
        x = torch.randn(3, 256, 27, 36, device="cuda")
        w2t = torch.randn(256, 256, device="cuda") * 0.05
        b2 = torch.randn(256, device="cuda")
        fin = torch.nn.Conv2d(256, 1, 1).cuda()
        z = ops.conv1x1_mfma(x, w2t, b2)
        torch.testing.assert_close(ops.conv1x1_mfma(x, w2t, b2, head_weight=fin.weight, head_bias=fin.bias), fin(torch.relu(z)),
                                   rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(ops.conv1x1_mfma(x, w2t, b2, head_weight=fin.weight), fin(torch.relu(z)) - fin.bias.view(1, 1, 1, 1),
                                   rtol=1e-4, atol=1e-4)
        # the whole 256-channel head: fast path (dw kernels, MFMA 1x1, fused output layer) vs the reference's literal chain
        head = M.DynamicSegHead(in_dim=103, embed_dim=256).cuda().eval()
        for m in head.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 2.0); m.weight.data.normal_(1, 0.1); m.bias.data.normal_(0, 0.1)
        xs = torch.randn(2, 103, 24, 30, device="cuda")
        ref = xs
        for layer in (head.layer1, head.layer2, head.layer3, head.layer4):
            ref = layer.relu2(layer.bn2(layer.conv2(layer.relu1(layer.bn1(layer.conv1(ref))))))
        ref = head.conv(ref)
        scale = float(ref.abs().max())
        torch.testing.assert_close(head(xs), ref, rtol=1e-3, atol=1e-3 * max(scale, 1.0))


@pytest.mark.gpu
def test_resident_weights_1x1_kernel_against_the_lds_weights_kernel():
    """r4 conv1x1_rw_kernel (weights in registers, activation through a ring of LDS stages, persistent over pixel ranges) against
    conv1x1_mfma_kernel: both are fp32 fmaf chains on v_mfma_f32_32x32x2_f32 + folded bias [+ ReLU] (one k-ascending chain
    there, two k-interleaved ones here), on whole and partial last tiles, 1..3 batch items, Cin = 32 .. 256, tiny planes (fewer tiles than CUs)."""
    import os
    import torch
    from cvpr2020_manet_amd import _lib, ops
    lib = _lib.load()
    os.environ["MANET_TUNING"] = "1"
    g = torch.Generator(device="cuda").manual_seed(11)
    try:
        for (B, cin, h, w, relu) in ((3, 256, 120, 214, False), (2, 256, 30, 54, True), (1, 32, 4, 16, False), (3, 64, 21, 36, True),
                                     (1, 128, 9, 12, False), (2, 96, 7, 20, True), (1, 256, 1, 4, False), (2, 256, 120, 214, True),
                                     (4, 64, 120, 214, False), (5, 32, 67, 92, True)):
            x = torch.randn(B, cin, h, w, generator=g, device="cuda")
            w2t = torch.randn(cin, 256, generator=g, device="cuda") * 0.1
            b2 = torch.randn(256, generator=g, device="cuda")
            _lib.check(lib.manet_tune_set(8, 1), "manet_tune_set")       # the LDS-weights kernel
            want = ops.conv1x1_mfma(x, w2t, b2, relu_out=relu)
            _lib.check(lib.manet_tune_set(8, -2 ** 31), "manet_tune_set")  # automatic: resident weights where it applies
            got = ops.conv1x1_mfma(x, w2t, b2, relu_out=relu)
            # (the resident-weights kernel sums even and odd k-steps in two chains: the same fp32 products, another summation
            # order -- a few ulp of the accumulated magnitude, not bit for bit)
            scale = float(x.abs().max() * w2t.abs().max()) * cin
            assert float((got - want).abs().max()) <= 4e-7 * scale, (B, cin, h, w)
            ref = torch.nn.functional.conv2d(x, w2t.t().reshape(256, cin, 1, 1).contiguous(), b2)
            torch.testing.assert_close(got, ref.relu() if relu else ref, rtol=2e-4, atol=2e-4)
            # r5: a group's range is cut in half-tile units (a tile at a range boundary is computed by two workgroups, one pixel
            # parity each); an output's bits do not depend on where the cuts fall -- one image alone is cut elsewhere
            for b in range(B if B > 1 else 0):
                assert torch.equal(ops.conv1x1_mfma(x[b:b + 1].contiguous(), w2t, b2, relu_out=relu)[0], got[b]), (B, cin, h, w, b)
            for form in (3, 4):  # whole tiles / half-tile units whatever the launcher would choose
                _lib.check(lib.manet_tune_set(8, form), "manet_tune_set")
                assert torch.equal(ops.conv1x1_mfma(x, w2t, b2, relu_out=relu), got), (B, cin, h, w, form)
    finally:
        lib.manet_tune_set(8, -2 ** 31)


@pytest.mark.gpu
def test_fuzz_head_and_mask_step_kernels():
    """tools/fuzz_head.py: random shapes through the depthwise kernel, both 1x1 kernels (any Cin, add term, fused output layer),
    head input assembly, the fused layer-1 launch, the resident-weights kernel on half-tile-cut planes, label resize and upsample + argmax against torch (5 100 cases ran clean when it was written, 2 500 with the r5 kinds)"""
    from tools import fuzz_head
    assert fuzz_head.run(200, 20200614) == 0


@pytest.mark.gpu
def test_three_piece_split_pointwise_is_fp32_class():
    """ops.conv1x1_split with a three-piece SplitWeight (hi + mid + lo = 24 significand bits, six products per pair on the bf16
    MFMA, fp32 accumulation): against an fp64 convolution its error is the fp32 kernel's class -- bound per output
    2^-21 * sum_k |x_k| |w_k| (the dropped products < 2^-24 each, the rest is fp32 accumulation over up to 256 terms) -- two orders below the
    two-piece kernel's 2^-15; same forms (relu_out, add, fused output layer, any Cin, partial tiles, non-finite inputs)."""
    import torch
    from cvpr2020_manet_amd import ops
    torch.manual_seed(6)
    with torch.no_grad():
        worst3, worst32 = 0.0, 0.0
        for (B, cin, h, w) in ((3, 256, 30, 54), (2, 100, 9, 12), (3, 3, 21, 36), (1, 1, 2, 2), (2, 17, 6, 10), (2, 33, 5, 52),
                               (3, 256, 120, 214)):
            x = torch.randn(B, cin, h, w, device="cuda") * 1.5
            w2t = torch.randn(cin, 256, device="cuda") * 0.08
            b2 = torch.randn(256, device="cuda")
            sw = ops.SplitWeight(w2t, pieces=3)
            assert sw.pieces == 3 and sw.packed.numel() == -(-cin // 16) * 24576
            w64 = w2t.t().reshape(256, cin, 1, 1).double()
            ref = torch.nn.functional.conv2d(x.double(), w64, b2.double())
            mag = torch.nn.functional.conv2d(x.double().abs(), w64.abs()) + b2.abs().double().view(1, -1, 1, 1)
            got = ops.conv1x1_split(x, sw, b2)
            err = (got.double() - ref).abs()
            assert bool((err <= 2.0 ** -21 * mag).all()), (cin, float((err / (2.0 ** -21 * mag)).max()))
            worst3 = max(worst3, float((err / mag).max()))
            worst32 = max(worst32, float(((ops.conv1x1_mfma(x, w2t, b2).double() - ref).abs() / mag).max()))
            assert torch.equal(ops.conv1x1_split(x, sw, b2, relu_out=True), torch.relu(got))
            add = torch.randn(1, 256, h, w, device="cuda")
            torch.testing.assert_close(ops.conv1x1_split(x, sw, b2, add=add, relu_out=True), torch.relu(got + add), rtol=1e-6, atol=1e-6)
            fin = torch.nn.Conv2d(256, 1, 1).cuda()
            torch.testing.assert_close(ops.conv1x1_split(x, sw, b2, head_weight=fin.weight, head_bias=fin.bias),
                                       fin(torch.relu(got)), rtol=1e-5, atol=1e-5)
        print("max relative error (of sum |x||w| + |b|): three-piece split %.3g, fp32 MFMA kernel %.3g" % (worst3, worst32))
        assert worst3 < 4 * max(worst32, 2.0 ** -24)  # the fp32 kernel's class
        x = torch.randn(1, 8, 4, 8, device="cuda")
        x[0, 3, 1, 2] = float("inf")
        x[0, 5, 2, 7] = float("nan")
        y = ops.conv1x1_split(x, ops.SplitWeight(torch.ones(8, 256, device="cuda"), pieces=3), torch.zeros(256, device="cuda"))
        bad = ~torch.isfinite(y[0, 0])
        assert bad.sum().item() == 2 and bool(bad[1, 2]) and bool(bad[2, 7])
        with pytest.raises(ValueError):
            ops.SplitWeight(torch.ones(8, 256, device="cuda"), pieces=4)
    # the module route: pointwise="split3" against the fp32 head
    from cvpr2020_manet_amd.networks import IntVOS as M
    torch.manual_seed(7)
    with torch.no_grad():
        head = M.DynamicSegHead(in_dim=103, embed_dim=256).cuda().eval()
        for m in head.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 2.0); m.weight.data.normal_(1, 0.1); m.bias.data.normal_(0, 0.1)
        xs = torch.randn(3, 103, 24, 30, device="cuda")
        outs = {}
        for mode in ("f32", "split3", "split"):
            for blk in head.modules():
                if isinstance(blk, M._split_separable_conv2d):
                    object.__setattr__(blk, "_pw_mode", mode)
            outs[mode] = head.forward_shared(xs[:1, :100].contiguous(), xs[:, 100:].contiguous())
        scale = float(outs["f32"].abs().max())
        d3, d2 = float((outs["split3"] - outs["f32"]).abs().max()), float((outs["split"] - outs["f32"]).abs().max())
        print("head logits vs the fp32 head: three-piece %.3g, two-piece %.3g (scale %.3g)" % (d3, d2, scale))
        assert d3 <= 2e-6 * max(scale, 1.0) and d3 < d2



@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 120, 214), (2, 24, 36), (5, 31, 44), (1, 8, 12), (6, 20, 68), (4, 10, 130)])
def test_fused_layer1_object_half_equals_the_three_launch_route(shape):
    """r5: head-input assembly + the per-object channels' depthwise 7x7 / bn1 / relu1 + their 1x1 / bn2 + the shared half's term +
    relu2 in ONE launch (manet_head_layer1_object_f32) against head_inputs -> dwconv7x7_bn_relu -> conv1x1_mfma(add=term): the same
    bits, ragged tile edges included; and against the literal module chain on the concatenated input"""
    from cvpr2020_manet_amd import ops
    from cvpr2020_manet_amd.networks import IntVOS as M
    n_ids, h, w = shape
    torch.manual_seed(h * w)
    head = M.DynamicSegHead(in_dim=103, embed_dim=256).cuda().eval()
    for m in head.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 2.0); m.weight.data.normal_(1, 0.2); m.bias.data.normal_(0, 0.2)
    emb = torch.relu(torch.randn(1, 100, h, w, device="cuda")) * 0.3
    gmap = torch.rand(h, w, n_ids, device="cuda")
    lmap = torch.rand(h, w, n_ids, device="cuda")
    lab = torch.randint(0, n_ids + 1, (h, w), device="cuda", dtype=torch.int32)  # (one label value matches no object)
    with torch.no_grad():
        fused = M._layer1_fused(head.layer1, emb, gmap, lmap, lab, n_ids, (h, w))
        assert fused is not None and tuple(fused.shape) == (n_ids, 256, h, w)
        per_object = ops.head_inputs(gmap, lmap, lab, n_ids, (h, w))
        three = head.layer1.forward_shared(emb, per_object)
        assert torch.equal(fused, three)
        lit = head.layer1(torch.cat((emb.repeat(n_ids, 1, 1, 1), per_object), 1))  # the general fast path on the concatenated input
        torch.testing.assert_close(fused, lit, rtol=2e-4, atol=2e-4)
        memo = {}
        a = M._layer1_fused(head.layer1, emb, gmap, lmap, lab, n_ids, (h, w), memo=memo)
        b = M._layer1_fused(head.layer1, emb, gmap, lmap, lab, n_ids, (h, w), memo=memo)   # second call: the memoised term
        assert torch.equal(a, fused) and torch.equal(b, fused) and memo["term"] is not None


@pytest.mark.gpu
def test_fused_layer1_refuses_what_it_cannot_take():
    """ops.head_layer1_object: operands on another device, a term of the wrong size or type, tensors that require grad -- errors,
    not launches"""
    from cvpr2020_manet_amd import ops
    h, w, n = 8, 12, 2
    dev = torch.device("cuda", 0)
    g, l = torch.rand(h, w, n, device=dev), torch.rand(h, w, n, device=dev)
    lab = torch.zeros(h, w, dtype=torch.int32, device=dev)
    wd, w2, b2 = torch.randn(3, 1, 7, 7, device=dev), torch.randn(3, 256, device=dev), torch.randn(256, device=dev)
    term = torch.randn(1, 256, h, w, device=dev)
    ok = ops.head_layer1_object(g, l, lab, n, (h, w), wd, None, None, None, w2, b2, term)
    assert tuple(ok.shape) == (n, 256, h, w)
    with pytest.raises(ValueError):
        ops.head_layer1_object(g, l.cpu(), lab, n, (h, w), wd, None, None, None, w2, b2, term)
    with pytest.raises(ValueError):
        ops.head_layer1_object(g, l, lab, n, (h, w), wd, None, None, None, w2, b2, term[:, :128])
    with pytest.raises(ValueError):
        ops.head_layer1_object(g, l, lab, n, (h, w), wd, None, None, None, w2.double(), b2, term)
    with pytest.raises(ValueError):
        ops.head_layer1_object(g, l, lab[:4], n, (h, w), wd, None, None, None, w2, b2, term)
    with pytest.raises(ValueError):
        ops.head_layer1_object(g, l, lab, n, (h, w), wd[:2], None, None, None, w2, b2, term)
    with pytest.raises(RuntimeError):
        ops.head_layer1_object(g.requires_grad_(), l, lab, n, (h, w), wd, None, None, None, w2, b2, term)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 256, 120, 214), (2, 5, 121, 150), (1, 3, 64, 129), (2, 4, 250, 80), (1, 2, 7, 66)])
def test_depthwise_narrow_last_column_tiles_equal_the_standard_tiling(shape):
    """r6: a plane's last tile column that holds at most 24 image columns (480p: 214 = 3 x 64 + 22) is covered by 120 x 24 tiles instead
    of 60 x 64 ones (7 workgroups per plane instead of 8).  Same taps in the same order: bit-equal to the standard tiling (tuning key 13
    = 0), on shapes with one / several narrow tiles per plane, ragged bottoms, even and odd widths, the per-item grid of few channels."""
    import os
    from cvpr2020_manet_amd import _lib, ops
    torch.manual_seed(sum(shape))
    B, C, h, w = shape
    x = torch.randn(B, C, h, w, device="cuda")
    wt = torch.randn(C, 1, 7, 7, device="cuda") * 0.2
    bias = torch.randn(C, device="cuda")
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    lib = _lib.load()
    os.environ["MANET_TUNING"] = "1"
    with torch.no_grad():
        got = ops.dwconv7x7_bn_relu(x, wt, bias, scale=sc, shift=sh, relu_in=True)
        _lib.check(lib.manet_tune_set(13, 0), "manet_tune_set")  # MANET_TUNE_DW_NARROW
        try:
            std = ops.dwconv7x7_bn_relu(x, wt, bias, scale=sc, shift=sh, relu_in=True)
        finally:
            _lib.check(lib.manet_tune_set(13, -2 ** 31), "manet_tune_set")
        want = torch.relu((torch.nn.functional.conv2d(torch.relu(x), wt, bias, padding=3, groups=C)) * sc[None, :, None, None]
                          + sh[None, :, None, None])
    assert torch.equal(got, std)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-4)
