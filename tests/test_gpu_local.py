"""GPU parity of the local-window matching and correlation HIP paths (through the C ABI) against the
committed reference vectors and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

# expf on the device vs glibc/torch differ in the last ulp; distances to fp32 rounding.
RTOL, ATOL = 1e-5, 2e-6


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cvpr2020_manet_amd import ops as o
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def chw_view(chw):
    return dev(chw).permute(1, 2, 0)


LOCAL_CASES = ["C16_12x15_d2", "C16_12x15_d4", "C100_10x14_d3", "C8_9x11_d1"]


@pytest.mark.parametrize("ds", [1, 0])
@pytest.mark.parametrize("case", LOCAL_CASES)
def test_golden_reference_vectors(ops, case, ds):
    g = load_golden("local_ds%d_%s" % (ds, case))
    d, n_ids = int(g["d"]), int(g["n_ids"])
    dist = ops.local_dist(chw_view(g["cur_chw"]), chw_view(g["prev_chw"]), d, downsample=bool(ds)).cpu().numpy()
    ref = g["dist"]
    assert np.array_equal(np.isinf(dist), np.isinf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(dist[fin], ref[fin], rtol=RTOL, atol=ATOL)
    out = ops.local_match(chw_view(g["prev_chw"]), chw_view(g["cur_chw"]), dev(g["labels"]), n_ids, d,
                          downsample=bool(ds)).cpu().numpy()
    np.testing.assert_allclose(out.reshape(g["out"].shape), g["out"], rtol=RTOL, atol=ATOL)


def _case(seed, h, w, C, n_ids, scale=0.1):
    rng = np.random.default_rng(seed)
    prev = (np.maximum(rng.standard_normal((C, h, w)), 0) * scale).astype(np.float32)
    cur = (np.maximum(rng.standard_normal((C, h, w)), 0) * scale).astype(np.float32)
    lab = rng.integers(-1, n_ids, size=(h, w, 1)).astype(np.int32)
    return prev, cur, lab


@pytest.mark.parametrize("shape", [
    (1, 31, 45, 100, 3, 4, 1),    # odd sizes -> pooled 15x22
    (2, 30, 54, 100, 2, 12, 1),   # reference default window (625 offsets)
    (3, 17, 19, 100, 10, 2, 1),   # more ids than one pass of the min kernel (8)
    (4, 21, 18, 64, 3, 3, 0),     # no downsample
    (5, 8, 8, 5, 2, 0, 1),        # d = 0 (single offset)
    (6, 2, 3, 7, 2, 1, 1),        # pooled grid 1x1
])
def test_vs_oracle(ops, oracle, shape):
    seed, h, w, C, n_ids, d, ds = shape
    prev, cur, lab = _case(seed, h, w, C, n_ids)
    out = ops.local_match(chw_view(prev), chw_view(cur), dev(lab), n_ids, d, downsample=bool(ds)).cpu().numpy()
    want = oracle.local_match(np.transpose(prev, (1, 2, 0)), np.transpose(cur, (1, 2, 0)), lab, n_ids, d,
                              downsample=bool(ds)).reshape(h, w, n_ids)
    np.testing.assert_allclose(out, want, rtol=RTOL, atol=ATOL)
    dist = ops.local_dist(chw_view(cur), chw_view(prev), d, downsample=bool(ds)).cpu().numpy()
    wd = oracle.local_dist(np.transpose(cur, (1, 2, 0)), np.transpose(prev, (1, 2, 0)), d, downsample=bool(ds))
    assert np.array_equal(np.isinf(dist), np.isinf(wd))
    fin = np.isfinite(wd)
    np.testing.assert_allclose(dist[fin], wd[fin], rtol=RTOL, atol=ATOL)


def test_row_major_inputs_give_the_same_result(ops):
    prev, cur, lab = _case(9, 24, 30, 100, 3)
    a = ops.local_match(chw_view(prev), chw_view(cur), dev(lab), 3, 4)
    b = ops.local_match(chw_view(prev).contiguous(), chw_view(cur).contiguous(), dev(lab), 3, 4)
    assert torch.equal(a, b)


def test_int_seghead_style_self_match(ops, oracle):
    """int_seghead matches a frame against itself with scribble labels incl. -1 (IntVOS.py:709-711)."""
    prev, _, lab = _case(10, 20, 26, 100, 3)
    out = ops.local_match(chw_view(prev), chw_view(prev), dev(lab), 3, 4).cpu().numpy()
    want = oracle.local_match(np.transpose(prev, (1, 2, 0)), np.transpose(prev, (1, 2, 0)), lab, 3, 4).reshape(20, 26, 3)
    np.testing.assert_allclose(out, want, rtol=RTOL, atol=ATOL)
    # a pixel whose own label is o has distance 0 to itself at the centre offset -> exactly 0
    own = lab[..., 0]
    for o in range(3):
        assert np.all(out[own == o, o] == 0.0)


def test_full_size_properties_cfg3(ops):
    """BASELINE cfg3: 120x214 grid, C=100, d=4, 4 ids."""
    torch.manual_seed(20200614 + 3)
    C, h, w, d, n_ids = 100, 120, 214, 4, 4
    prev = torch.relu(torch.randn(C, h, w, device="cuda")) * 0.1
    cur = torch.relu(torch.randn(C, h, w, device="cuda")) * 0.1
    lab = torch.zeros(h, w, dtype=torch.int32, device="cuda")
    lab[20:60, 30:90] = 1; lab[70:100, 100:180] = 2; lab[5:15, 150:200] = 3
    out = ops.local_match(prev.permute(1, 2, 0), cur.permute(1, 2, 0), lab, n_ids, d)
    assert out.shape == (h, w, n_ids)
    assert out.min().item() >= 0.0 and out.max().item() <= 1.0  # normalised distances
    # far from object 3 (further than 2d pixels) nothing can match it -> exactly 1.0
    far = torch.ones(h, w, dtype=torch.bool, device="cuda"); far[0:15 + 2 * d + 1, 150 - 2 * d:200 + 2 * d + 1] = False
    assert torch.all(out[..., 3][far] == 1.0)
    # self match: distance 0 for the pixel's own label
    same = ops.local_match(cur.permute(1, 2, 0), cur.permute(1, 2, 0), lab, n_ids, d)
    assert torch.all(same.gather(2, lab.long()[..., None]) == 0.0)
    # the fused path equals masked-min over the materialised volume (IntVOS.py:428-432)
    vol = ops.local_dist(cur.permute(1, 2, 0), prev.permute(1, 2, 0), d)  # [h,w,81]
    P = 2 * d + 1
    padl = torch.nn.functional.pad(lab.float()[None, None], (2 * d,) * 4)
    offl = torch.nn.functional.unfold(padl, kernel_size=(h, w), stride=(2, 2)).view(h, w, P * P, 1)
    mask = offl == torch.arange(n_ids, device="cuda").float()
    want = torch.where(mask, vol[..., None].expand(-1, -1, -1, n_ids), torch.ones((), device="cuda")).min(dim=2).values
    assert torch.equal(out, want)


def test_correlation_vs_oracle(ops, oracle):
    rng = np.random.default_rng(3)
    for (B, C, H, W, pad, K, md, s1, s2) in [(2, 5, 9, 8, 3, 3, 2, 2, 2), (1, 100, 20, 27, 4, 1, 4, 1, 1),
                                              (1, 7, 12, 12, 6, 1, 6, 1, 2)]:
        a = rng.standard_normal((B, C, H, W)).astype(np.float32)
        b = rng.standard_normal((B, C, H, W)).astype(np.float32)
        out = ops.correlation_forward(dev(a), dev(b), pad, K, md, s1, s2).cpu().numpy()
        want = oracle.correlation_forward(a, b, pad, K, md, s1, s2)
        assert out.shape == want.shape
        if K == 1:
            np.testing.assert_array_equal(out, want)  # same fmaf chain order
        else:  # the LDS-tiled kernel sums (channel chunk, j, i, c) instead of (j, i, c): fp32 rounding
            np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float64])
@pytest.mark.parametrize("cfgc", [(2, 5, 9, 8, 3, 3, 2, 2, 2), (1, 100, 20, 27, 4, 1, 4, 1, 1), (1, 16, 14, 40, 12, 1, 12, 1, 1)])
def test_correlation_half_and_double(ops, dtype, cfgc):
    """the reference's forward dispatches float / double / half (correlation_cuda_kernel.cu:386-415): product in the
    tensor's type, fp32 accumulation, mean stored in the tensor's type (:121-143)"""
    B, C, H, W, pad, K, md, s1, s2 = cfgc
    g = torch.Generator().manual_seed(5)
    a = (torch.randn(B, C, H, W, generator=g) * 0.5).to(dtype).cuda()
    b = (torch.randn(B, C, H, W, generator=g) * 0.5).to(dtype).cuda()
    out = ops.correlation_forward(a, b, pad, K, md, s1, s2)
    assert out.dtype == dtype
    ref = ops.correlation_forward(a.float(), b.float(), pad, K, md, s1, s2)  # fp32 kernel on the same values
    if dtype == torch.float16:
        # each product carries a 2^-11 relative rounding before the fp32 sum, the result one more
        scale = (a.float().abs().max() * b.float().abs().max()).item()
        assert (out.float() - ref).abs().max().item() < 3e-3 * scale
    else:
        torch.testing.assert_close(out.float(), ref, rtol=1e-5, atol=1e-6)


def test_correlation_tied_to_reference_distance(ops):
    g = load_golden("correlation_tie")
    a, b, d = g["in1"], g["in2"], int(g["d"])
    C, h, w = a.shape
    corr = ops.correlation_forward(dev(a[None]), dev(b[None]), d, 1, d, 1, 1).cpu().numpy()
    xs = (a.astype(np.float64) ** 2).sum(0)
    ys = (b.astype(np.float64) ** 2).sum(0)
    P = 2 * d + 1
    for dy in range(P):
        for dx in range(P):
            for y in range(h):
                for x in range(w):
                    yy, xx = y + dy - d, x + dx - d
                    r = g["dist"][y, x, dy * P + dx]
                    if 0 <= yy < h and 0 <= xx < w:
                        assert abs(xs[y, x] + ys[yy, xx] - 2 * C * corr[0, dy * P + dx, y, x] - r) < 1e-4 * max(1.0, r)
                    else:
                        assert corr[0, dy * P + dx, y, x] == 0.0


@pytest.mark.parametrize("d", range(13))
def test_fused_kernel_every_window_size(ops, oracle, d):
    """Every instantiation of the fused kernel (d = 0..12: different thread maps, window halves, LDS strides, 1-5
    workgroups per tile) on a grid with several tile rows AND columns, ragged edges, C spanning several LDS stages with a
    partial last one, more ids than one per-pixel pass holds, labels outside [0, n_ids): against the oracle within the
    local tolerance, and bit for bit against the three-launch path over the materialised volume (tuning key 4)."""
    from cvpr2020_manet_amd import _lib
    rng = np.random.default_rng(4200 + d)
    for (C, h, w, n_ids) in ((58, 53, 71, 11), (7, 9, 100, 2)):
        prev = (np.maximum(rng.standard_normal((C, h, w)), 0) * 0.2).astype(np.float32)
        cur = (np.maximum(rng.standard_normal((C, h, w)), 0) * 0.2).astype(np.float32)
        lab = rng.integers(-1, n_ids + 1, size=(h, w, 1)).astype(np.int32)
        p, c, l = chw_view(prev), chw_view(cur), dev(lab)
        got = ops.local_match(p, c, l, n_ids, d)
        want = oracle.local_match(np.transpose(prev, (1, 2, 0)), np.transpose(cur, (1, 2, 0)), lab, n_ids, d,
                                  downsample=True).reshape(h, w, n_ids)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=RTOL, atol=ATOL)
        lib = _lib.load()
        os.environ["MANET_TUNING"] = "1"  # the setters refuse without the opt-in
        _lib.check(lib.manet_tune_set(4, 1), "manet_tune_set")  # MANET_TUNE_LOCAL_UNFUSED
        try:
            unfused = ops.local_match(p, c, l, n_ids, d)
        finally:
            _lib.check(lib.manet_tune_set(4, 0), "manet_tune_set")
        assert torch.equal(got, unfused)
        # 2-byte embeddings: the pooling pass reads bf16, everything after it is the same fp32 arithmetic
        pb, cb = p.to(torch.bfloat16), c.to(torch.bfloat16)
        got_b = ops.local_match(pb, cb, l, n_ids, d)
        ref_b = ops.local_match(pb.float(), cb.float(), l, n_ids, d)
        assert torch.equal(got_b, ref_b)


def test_full_size_local_cfg5_grid(ops, oracle):
    """BASELINE cfg5's grid (720p: 180x320, C=100, d=4, 6 ids), 2-byte embeddings as bench.py runs it: the fused kernel
    against the oracle on the same (bf16-rounded) inputs, and bit for bit against the materialised-volume path."""
    from cvpr2020_manet_amd import _lib
    torch.manual_seed(20200614 + 5)
    C, h, w, d, n_ids = 100, 180, 320, 4, 6
    prev = (torch.relu(torch.randn(C, h, w, device="cuda")) * 0.1).to(torch.bfloat16)
    cur = (torch.relu(torch.randn(C, h, w, device="cuda")) * 0.1).to(torch.bfloat16)
    lab = torch.randint(0, n_ids, (h, w), dtype=torch.int32, device="cuda")
    got = ops.local_match(prev.permute(1, 2, 0), cur.permute(1, 2, 0), lab, n_ids, d)
    pf, cf = prev.float(), cur.float()
    want = oracle.local_match(pf.permute(1, 2, 0).cpu().numpy(), cf.permute(1, 2, 0).cpu().numpy(),
                              lab.cpu().numpy().reshape(h, w, 1), n_ids, d, downsample=True).reshape(h, w, n_ids)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=RTOL, atol=ATOL)
    lib = _lib.load()
    os.environ["MANET_TUNING"] = "1"
    _lib.check(lib.manet_tune_set(4, 1), "manet_tune_set")  # MANET_TUNE_LOCAL_UNFUSED
    try:
        unfused = ops.local_match(pf.permute(1, 2, 0), cf.permute(1, 2, 0), lab, n_ids, d)
    finally:
        _lib.check(lib.manet_tune_set(4, 0), "manet_tune_set")
    assert torch.equal(got, unfused)


@pytest.mark.parametrize("d", range(13))
def test_stored_volume_path_every_window_size(ops, oracle, d):
    """r6: phase 1 (window distances, label-independent) batched into stored volumes, phase 2 (label gather + masked min) on a
    stored volume -- every instantiation d = 0..12, ragged grids, more ids than one per-pixel pass holds, fp32 and 2-byte
    embeddings: bit for bit the fused kernel's result (which the test above ties to the oracle), and against the oracle itself."""
    rng = np.random.default_rng(6200 + d)
    for (C, h, w, n_ids, dtype) in ((58, 53, 71, 11, torch.float32), (7, 9, 100, 2, torch.float32),
                                    (33, 30, 54, 3, torch.bfloat16)):
        embs = torch.from_numpy((np.maximum(rng.standard_normal((4, C, h, w)), 0) * 0.2).astype(np.float32)).cuda().to(dtype)
        labs = torch.from_numpy(rng.integers(-1, n_ids + 1, size=(4, h, w)).astype(np.int32)).cuda()
        frames = ops.prepare_frames(embs, compute="f32", max_distance=d)
        # forward pairs (t-1 -> t), backward pairs (t+1 -> t), a frame against itself (int_seghead, IntVOS.py:709-711)
        pairs = [(0, 1), (1, 2), (2, 3), (3, 2), (2, 1), (1, 0), (2, 2)]
        vols = ops.local_volumes([frames[a] for a, _ in pairs], [frames[b] for _, b in pairs])
        assert vols.shape[0] == len(pairs) and vols.shape[1] * 4 == ops.local_volume_bytes(h, w, d)
        for i, (a, b) in enumerate(pairs):
            want = ops.local_match_frames(frames[a], frames[b], labs[a], n_ids)
            got = ops.local_match_volume(vols[i], frames[b], labs[a], n_ids)
            assert torch.equal(got, want), (d, a, b)
            # a pre-set output buffer, as prop_seghead hands it over
            pre = torch.ones((h, w, n_ids), dtype=torch.float32, device="cuda")
            got2 = ops.local_match_volume(vols[i], frames[b], labs[a], n_ids, out=pre, out_is_preset=True)
            assert torch.equal(got2, want)
        a, b = pairs[0]
        e = embs.float().cpu().numpy()
        ref = oracle.local_match(np.transpose(e[a], (1, 2, 0)), np.transpose(e[b], (1, 2, 0)),
                                 labs[a].cpu().numpy().reshape(h, w, 1), n_ids, d, downsample=True).reshape(h, w, n_ids)
        np.testing.assert_allclose(ops.local_match_volume(vols[0], frames[b], labs[a], n_ids).cpu().numpy(), ref, rtol=RTOL, atol=ATOL)


def test_stored_volumes_more_pairs_than_one_launch(ops):
    """70 frame pairs = three launches of the batched phase-1 kernel (32 pairs each); every volume equals the one made alone;
    the labels change between uses of a volume (interaction rounds), the volume does not."""
    torch.manual_seed(66)
    C, h, w, d, n_ids = 24, 22, 38, 12, 3
    embs = torch.relu(torch.randn(36, C, h, w, device="cuda")) * 0.2
    frames = ops.prepare_frames(embs, compute="f32", max_distance=d)
    pairs = [(t - 1, t) for t in range(1, 36)] + [(t + 1, t) for t in range(35)]
    vols = ops.local_volumes([frames[a] for a, _ in pairs], [frames[b] for _, b in pairs])
    for i in (0, 31, 32, 63, 64, 69):
        a, b = pairs[i]
        alone = ops.local_volumes([frames[a]], [frames[b]])
        for rnd in range(2):
            lab = torch.randint(0, n_ids, (h, w), dtype=torch.int32, device="cuda")
            want = ops.local_match_frames(frames[a], frames[b], lab, n_ids)
            assert torch.equal(ops.local_match_volume(vols[i], frames[b], lab, n_ids), want)
            assert torch.equal(ops.local_match_volume(alone[0], frames[b], lab, n_ids), want)


def test_stored_volume_full_size_480p(ops):
    """BASELINE cfg2's grid at the reference's default window (120x214, C=100, d=12, 3 ids): bit-equal to the fused kernel"""
    torch.manual_seed(20200614 + 2)
    C, h, w, d, n_ids = 100, 120, 214, 12, 3
    embs = torch.relu(torch.randn(2, C, h, w, device="cuda")) * 0.1
    lab = torch.randint(0, n_ids, (h, w), dtype=torch.int32, device="cuda")
    frames = ops.prepare_frames(embs, compute="f32", max_distance=d)
    vols = ops.local_volumes([frames[0]], [frames[1]])
    assert vols.numel() * 4 == ops.local_volume_bytes(h, w, d) == 240 * 107520 + 1024  # (+ the last LDS-DMA piece's slack)
    assert torch.equal(ops.local_match_volume(vols[0], frames[1], lab, n_ids), ops.local_match_frames(frames[0], frames[1], lab, n_ids))


@pytest.mark.parametrize("d,n_ids", [(4, 6), (12, 6), (12, 11)])
def test_stored_volume_full_size_720p_grid(ops, d, n_ids):
    """BASELINE configs[4]'s grid (180x320, C = 100) on 2-byte embeddings: two rounds of workgroups per launch at d = 12, more ids than
    one per-pixel pass holds (11), labels outside [0, n_ids): the stored-volume path bit-equal to the fused kernel, both directions"""
    torch.manual_seed(20200614 + 5 + d)
    C, h, w = 100, 180, 320
    embs = (torch.relu(torch.randn(3, C, h, w, device="cuda")) * 0.1).to(torch.bfloat16)
    lab = torch.randint(-1, n_ids + 1, (h, w), dtype=torch.int32, device="cuda")
    lab[40:120, 60:200] = 1  # a mask-like region too: rows whose window sees one label take the register path
    frames = ops.prepare_frames(embs, compute="bf16", max_distance=d)
    pairs = [(0, 1), (1, 2), (2, 1), (1, 0)]
    vols = ops.local_volumes([frames[a] for a, _ in pairs], [frames[b] for _, b in pairs])
    for i, (a, b) in enumerate(pairs):
        want = ops.local_match_frames(frames[a], frames[b], lab, n_ids)
        assert torch.equal(ops.local_match_volume(vols[i], frames[b], lab, n_ids), want)
