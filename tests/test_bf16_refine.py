"""MANET_COMPUTE_BF16_REFINE ("bf16r"): bf16 filter + exact fp32 re-rank (VERDICT r2 "next" #4).  IntVOS.py:81-85 is a
minimum over the bank, so a filter that provably keeps the arg-min row is exact: the result must EQUAL the fp32 MFMA
kernel's (which is bit-exact against the pinned oracle, test_gpu_global.py) bit for bit -- on random sweeps, scribble-like
banks, empty objects, larger embedding scales (where plain bf16 leaves the 1e-3 bar), stacked multi-frame banks, prepared
frames, the fused epilogue, and on adversarial banks of duplicated rows that overflow every candidate list."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cvpr2020_manet_amd import ops as o
    return o


def _case(seed, N, M, C, n_ids, scale=0.2, unlabelled=0.0, dtype=torch.float32):
    g = torch.Generator(device="cuda").manual_seed(seed)
    q = (torch.relu(torch.randn(N, C, generator=g, device="cuda")) * scale).to(dtype)
    k = (torch.relu(torch.randn(M, C, generator=g, device="cuda")) * scale).to(dtype)
    lab = torch.randint(0, n_ids, (M,), generator=g, device="cuda", dtype=torch.int32)
    if unlabelled > 0:
        lab[torch.rand(M, generator=g, device="cuda") < unlabelled] = -1
    return q, k, lab


@pytest.mark.parametrize("shape", [(700, 3000, 100, 3), (1, 64, 100, 1), (257, 65, 16, 2), (3000, 20000, 100, 5),
                                   (512, 5000, 106, 4), (900, 900, 7, 2)])
@pytest.mark.parametrize("scale", [0.1, 0.5, 3.0])
def test_bit_equal_to_the_fp32_kernel(ops, shape, scale):
    N, M, C, n_ids = shape
    q, k, lab = _case(N + M, N, M, C, n_ids, scale)
    want = ops.global_match(k, q, lab, n_ids, compute="f32")
    got = ops.global_match(k, q, lab, n_ids, compute="bf16r")
    assert torch.equal(got, want)


def test_scribble_banks_empty_objects_and_the_fused_epilogue(ops, oracle):
    N, M, C, n_ids = 1500, 6000, 100, 5
    q, k, lab = _case(5, N, M, C, n_ids, 0.2, unlabelled=0.9)
    lab[lab == 3] = -1  # object 3 has no row at all -> the padding distance 1e20 -> 1.0 after normalisation
    mem_a, mem_b = torch.full((N, n_ids), 0.4, device="cuda"), torch.full((N, n_ids), 0.4, device="cuda")
    want = ops.global_match(k, q, lab, n_ids, compute="f32", normalize=True, mem=mem_a)
    got = ops.global_match(k, q, lab, n_ids, compute="bf16r", normalize=True, mem=mem_b)
    assert torch.equal(got, want) and torch.equal(mem_a, mem_b)
    raw = ops.global_match(k, q, lab, n_ids, compute="bf16r")
    assert bool((raw[:, 3] == 1e20).all())
    # and against the oracle itself
    ref = oracle.global_match(k.cpu().numpy().reshape(M, 1, C), q[:200].cpu().numpy().reshape(200, 1, C),
                              lab.cpu().numpy().reshape(M, 1, 1), 1, n_ids=n_ids).reshape(200, n_ids)
    assert np.array_equal(raw[:200].cpu().numpy(), ref)
    # an all-unlabelled bank
    lab[:] = -1
    assert bool((ops.global_match(k, q, lab, n_ids, compute="bf16r") == 1e20).all())


def test_prepared_bank_frames_and_2_byte_storage(ops):
    C, h, w, n_ids = 100, 40, 50, 3
    g = torch.Generator(device="cuda").manual_seed(8)
    emb = torch.relu(torch.randn(3, C, h, w, generator=g, device="cuda")) * 0.3
    lab = torch.randint(-1, n_ids, (2 * h * w,), generator=g, device="cuda", dtype=torch.int32)
    for storage in (torch.float32, torch.bfloat16):
        e = emb.to(storage)
        bank_rows = e[:2].permute(0, 2, 3, 1).reshape(-1, C)  # a stacked 2-frame bank
        ref_bank = ops.PreparedBank(bank_rows, lab, n_ids, compute="f32")
        bank = ops.PreparedBank(bank_rows, lab, n_ids, compute="bf16r")
        want = ref_bank.match(e[2].permute(1, 2, 0))
        assert torch.equal(bank.match(e[2].permute(1, 2, 0)), want)
        for prep in ("bf16r", "bf16"):  # a frame prepared for plain bf16 carries the same operand image
            fr = ops.prepare_frames(e[2], compute=prep, max_distance=4)
            assert torch.equal(bank.match(fr), want)
        assert torch.equal(bank.match(ops.PackedQuery(e[2].permute(1, 2, 0), compute="bf16r")), want)
        cands, over = bank.refine_stats()
        assert over == 0 and h * w <= cands < 16 * h * w * n_ids


def test_adversarial_duplicate_rows_dense_blocks_and_beyond(ops):
    """banks that are many copies of a handful of rows: whole 32-query x 32-row blocks qualify.  (a) 40 copies: every block
    is listed as ONE dense entry and re-ranked exactly (r4; before, such a block sent its 256-query tile to the rescue pass);
    (b) 11 000 copies: more dense blocks per 32-query bucket than REFINE_DENSE_CAP (1 024) -- the buckets are marked incomplete
    and the rescue pass (the exact fp32 kernel) takes every tile.  Either way the fp32 kernel's bits."""
    N, C, n_ids = 300, 100, 2
    g = torch.Generator(device="cuda").manual_seed(21)
    q = torch.relu(torch.randn(N, C, generator=g, device="cuda")) * 0.2
    base = torch.relu(torch.randn(6, C, generator=g, device="cuda")) * 0.2
    for copies, rescued in ((40, False), (2200, False), (11000, True)):
        k = base.repeat(copies, 1)
        lab = (torch.arange(6 * copies, device="cuda") % 6 < 3).to(torch.int32)
        want = ops.global_match(k, q, lab, n_ids, compute="f32")
        bank = ops.PreparedBank(k, lab, n_ids, compute="bf16r")
        got = bank.match(q, adaptive=False)
        assert torch.equal(got, want)
        st = bank.refine_stats_full()
        assert st["candidate_rows"] > 0  # (listed rows + per dense block the lanes that held a qualifying row: a lower bound)
        if rescued:
            assert st["list_overflowed"] == 1 and st["rescued_tiles"] == st["query_tiles"] > 0
        else:
            assert st["list_overflowed"] == 0 and st["rescued_tiles"] == 0


def test_dense_blocks_with_nan_and_2_byte_storage(ops):
    """dense entries take the same chains as listed rows: NaN rows / queries propagate as in the fp32 kernel, bf16-stored
    embeddings re-rank from the stored values"""
    N, C, n_ids = 200, 100, 3
    g = torch.Generator(device="cuda").manual_seed(5)
    base = torch.relu(torch.randn(4, C, generator=g, device="cuda")) * 0.2
    k = (base.repeat(100, 1) + 1e-4 * torch.randn(400, C, generator=g, device="cuda")).contiguous()
    lab = (torch.arange(400, device="cuda") % 3).to(torch.int32)
    q = (base[torch.randint(0, 4, (N,), generator=g, device="cuda")] + 1e-4 * torch.randn(N, C, generator=g, device="cuda")).contiguous()
    for storage in (torch.float32, torch.bfloat16):
        ks, qs = k.to(storage), q.to(storage)
        want = ops.global_match(ks, qs, lab, n_ids, compute="f32")
        bank = ops.PreparedBank(ks, lab, n_ids, compute="bf16r")
        assert torch.equal(bank.match(qs, adaptive=False), want)
        assert bank.refine_stats_full()["rescued_tiles"] == 0
    kn = k.clone()
    kn[7, 3] = float("nan")  # object 7 % 3 == 1
    qn = q.clone()
    qn[11, 0] = float("nan")
    want = ops.global_match(kn, qn, lab, n_ids, compute="f32")
    got = ops.PreparedBank(kn, lab, n_ids, compute="bf16r").match(qn, adaptive=False)
    assert torch.equal(torch.isnan(got), torch.isnan(want)) and torch.equal(torch.nan_to_num(got, nan=-1.0), torch.nan_to_num(want, nan=-1.0))
    assert bool(torch.isnan(got[:, 1]).all()) and bool(torch.isnan(got[11]).all())


@pytest.mark.parametrize("C", [4, 7, 16, 26, 30, 50, 99, 101, 106])
@pytest.mark.parametrize("storage", [torch.float32, torch.bfloat16])
def test_dense_blocks_over_channel_counts_and_ragged_sizes(ops, C, storage):
    """the dense re-rank's 32 x 32 x C matrix tile: C a multiple of 4 (16-byte row reads) or not (the generic k loop), the
    small-C filter kernel (C <= 26) and the C = 100-class one, query counts that are not multiples of 32 / 256, a bank whose
    last tile is partly padding; near-duplicate rows so that whole blocks qualify"""
    N, n_ids = 333, 3
    g = torch.Generator(device="cuda").manual_seed(1000 + C)
    base = torch.relu(torch.randn(5, C, generator=g, device="cuda")) * 0.3 + 0.05
    M = 5 * 131  # 655 rows: objects end inside tiles
    k = (base.repeat(131, 1) + 2e-4 * torch.randn(M, C, generator=g, device="cuda")).to(storage)
    lab = (torch.arange(M, device="cuda") % n_ids).to(torch.int32)
    q = (base[torch.randint(0, 5, (N,), generator=g, device="cuda")] + 2e-4 * torch.randn(N, C, generator=g, device="cuda")).to(storage)
    want = ops.global_match(k, q, lab, n_ids, compute="f32")
    bank = ops.PreparedBank(k, lab, n_ids, compute="bf16r")
    got = bank.match(q, adaptive=False)
    assert torch.equal(got, want), (C, storage)
    st = bank.refine_stats_full()
    assert st["rescued_tiles"] == 0 and st["candidate_rows_per_pair"] > 2.0  # (whole blocks qualified: dense entries)


@pytest.mark.parametrize("cfg", [3, 5])
def test_full_size_bit_equal_and_candidate_count(ops, cfg):
    H, W, T, n_ids = {3: (120, 214, 5, 4), 5: (180, 320, 10, 6)}[cfg]
    g = torch.Generator(device="cuda").manual_seed(20200614 + cfg)
    cur = torch.relu(torch.randn(100, H, W, generator=g, device="cuda")) * 0.1
    bank_rows = torch.relu(torch.randn(T * H * W, 100, generator=g, device="cuda")) * 0.1
    lab = torch.randint(0, n_ids, (T * H * W,), generator=g, device="cuda", dtype=torch.int32)
    want = ops.global_match(bank_rows, cur.permute(1, 2, 0), lab, n_ids, compute="f32")
    bank = ops.PreparedBank(bank_rows, lab, n_ids, compute="bf16r")
    got = bank.match(cur.permute(1, 2, 0))
    assert torch.equal(got, want)
    cands, over = bank.refine_stats()
    per_pair = cands / (H * W * n_ids)
    print("cfg%d: %.2f candidate rows per (query, object), list overflowed: %d" % (cfg, per_pair, over))
    assert over == 0 and 1.0 <= per_pair < 12.0


def test_errors_are_loud(ops):
    q, k, lab = _case(1, 100, 500, 120, 2)  # C = 120 needs the narrow bf16 kernel: not offered in this mode
    with pytest.raises(RuntimeError, match="C <= 106"):
        ops.global_match(k, q, lab, 2, compute="bf16r")
    q, k, lab = _case(1, 100, 500, 100, 2)
    with pytest.raises(RuntimeError, match="k_nn"):
        ops.global_match(k, q, lab, 2, compute="bf16r", k_nearest_neighbors=2)


def test_spatially_smooth_embeddings_do_not_overflow(ops):
    """embeddings that vary slowly over the image (what a trained encoder produces; the random fields of the other tests are
    the easy case): whole 32-query blocks qualify together while the thresholds are still loose.  The filter pass must list
    them all (sub-lists are flushed to their buckets when full, nothing is dropped) -- r3's first form raised the overflow flag
    here and every pair fell back to a 0.8 s scan."""
    import torch.nn.functional as F
    H, W, T, n_ids = 60, 107, 4, 3
    g = torch.Generator(device="cuda").manual_seed(8)
    for coarse, noise in ((8, 0.1), (16, 0.0)):
        base = torch.randn(1, 100, H // coarse + 2, W // coarse + 2, generator=g, device="cuda")
        frames = []
        for i in range(T + 1):
            f = F.interpolate(base + 0.05 * i * torch.randn(base.shape, generator=g, device="cuda"), size=(H, W),
                              mode="bilinear", align_corners=True)[0]
            frames.append(torch.relu(f + noise * torch.randn(100, H, W, generator=g, device="cuda")) * 0.1)
        bank_rows = torch.stack(frames[:T]).permute(0, 2, 3, 1).reshape(-1, 100)
        lab = torch.randint(0, n_ids, (T * H * W,), generator=g, device="cuda", dtype=torch.int32)
        q = frames[T].permute(1, 2, 0)
        want = ops.global_match(bank_rows, q, lab, n_ids, compute="f32")
        bank = ops.PreparedBank(bank_rows, lab, n_ids, compute="bf16r")
        assert torch.equal(bank.match(q), want)
        cands, over = bank.refine_stats()
        assert over == 0 and cands / (H * W * n_ids) < 16.0


def test_partly_degenerate_embeddings_rescue_only_their_blocks(ops):
    """a band of the image where every embedding is the SAME vector (nothing for the bf16 pass to tell apart: hundreds of
    bank rows tie for every query of the band).  (a) a band of 25 bank rows x 64 columns: 50 dense 32 x 32 blocks per
    bucket -- re-ranked exactly, no rescue (r4); (b) the same vector in every row of 15 more bank frames: more dense blocks than a
    bucket takes (1 024) -- the band's query tiles, and only those, go through the rescue pass (the exact fp32 kernel, dealt to
    the listed tiles); the rest of the frame goes through the candidate lists.  The result is the fp32 kernel's, bit for bit."""
    H, W, n_ids = 40, 64, 2
    g = torch.Generator(device="cuda").manual_seed(9)
    q = torch.relu(torch.randn(100, H, W, generator=g, device="cuda")) * 0.2
    const = torch.relu(torch.randn(100, generator=g, device="cuda")) * 0.2
    q[:, 10:20, :] = const[:, None, None]
    for frames, rescued in ((1, False), (16, True)):
        ref = torch.relu(torch.randn(frames, 100, H, W, generator=g, device="cuda")) * 0.2
        ref[0, :, 5:30, :] = (const * 1.001)[:, None, None]
        ref[1:] = (const * 1.001)[None, :, None, None]  # (the other queries find their nearest rows in frame 0's random part)
        lab = torch.randint(0, n_ids, (frames * H * W,), generator=g, device="cuda", dtype=torch.int32)
        bank_rows = ref.permute(0, 2, 3, 1).reshape(-1, 100)
        want = ops.global_match(bank_rows, q.permute(1, 2, 0), lab, n_ids, compute="f32")
        bank = ops.PreparedBank(bank_rows, lab, n_ids, compute="bf16r")
        assert torch.equal(bank.match(q.permute(1, 2, 0), adaptive=False), want)
        st = bank.refine_stats_full()
        if rescued:  # queries 640 .. 1279 are the band: tiles 2, 3, 4 of the 10
            assert st["list_overflowed"] == 1 and 0 < st["rescued_tiles"] <= 4 and st["query_tiles"] == 10
        else:
            assert st["list_overflowed"] == 0 and st["rescued_tiles"] == 0


def _same_with_nans(a, b):
    na, nb = torch.isnan(a), torch.isnan(b)
    return bool(torch.equal(na, nb)) and bool(torch.equal(torch.where(na, torch.zeros_like(a), a),
                                                          torch.where(nb, torch.zeros_like(b), b)))


@pytest.mark.parametrize("where", ["bank_row_in_prepass_sample", "bank_row_outside_sample", "query_row", "both"])
def test_nan_embeddings_propagate_like_the_fp32_kernel(ops, where):
    """one NaN bank row used to turn max |k|^2 -- hence EVERY pair's threshold -- into NaN: no candidates, no
    rescue, the whole map came out as the padding distance.  Now: a NaN row poisons its own object only (as in the fp32 kernel,
    whose NaN-propagating minimum returns NaN for every query of that object), a NaN query row its own pixel only, and every
    other pair is still the fp32 kernel's bits."""
    N, M, C, n_ids = 700, 9000, 100, 3
    q, k, lab = _case(77, N, M, C, n_ids, 0.2)
    if where in ("bank_row_in_prepass_sample", "both"):
        k[int(torch.nonzero(lab == 1)[0])] = float("nan")       # first row of object 1: tile 0 of the object, always sampled
    if where == "bank_row_outside_sample":
        k[int(torch.nonzero(lab == 1)[200])] = float("nan")     # a row of the object's 4th tile: not in the 1-in-8 sample
    if where in ("query_row", "both"):
        q[123, 7] = float("nan")
    want = ops.global_match(k, q, lab, n_ids, compute="f32")
    got = ops.global_match(k, q, lab, n_ids, compute="bf16r")
    assert _same_with_nans(got, want)
    if where != "query_row":
        clean = torch.ones(N, dtype=torch.bool, device="cuda")
        clean[123] = where != "both"
        assert bool(torch.isnan(want[:, 1]).all()) and not bool(torch.isnan(want[clean][:, 0]).any())
    if where != "bank_row_in_prepass_sample" and where != "bank_row_outside_sample":
        assert bool(torch.isnan(want[123]).all())
    # normalised + merged epilogue on top (NaN stays NaN through sigmoid; `g <= mem` is false for NaN: the stored value wins)
    mem_a, mem_b = torch.full((N, n_ids), 0.4, device="cuda"), torch.full((N, n_ids), 0.4, device="cuda")
    a = ops.global_match(k, q, lab, n_ids, compute="f32", normalize=True, mem=mem_a)
    b = ops.global_match(k, q, lab, n_ids, compute="bf16r", normalize=True, mem=mem_b)
    assert _same_with_nans(a, b) and _same_with_nans(mem_a, mem_b)


def test_fuzz_structured_banks(ops):
    """tools/fuzz_bf16r.py: random mixtures of i.i.d. rows, near-duplicates, smooth ramps, constant bands, unlabelled rows, empty
    objects, NaN rows, 2-byte storage over random sizes / channel counts / id counts -- listed rows, dense blocks, rescued
    tiles all occur; every case must equal the fp32 kernel bit for bit (6 400 cases of it ran clean when the dense form was built)"""
    from tools import fuzz_bf16r
    assert fuzz_bf16r.run(150, 20200614, verbose=False) == 0
