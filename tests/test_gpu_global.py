"""GPU parity of the global matching HIP path (through the C ABI) against
  (1) the committed golden vectors produced by the reference itself, and
  (2) the CPU oracle on seeded inputs -- BIT-EXACT for the fp32 path (integer-like bar: the fp32
      MFMA is a k-ascending fmaf chain, which is how the oracle evaluates the reference's formula),
  (3) size-independent properties at BASELINE.json's full sizes."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

# vs the reference's own outputs (different summation order inside torch.matmul): fp32 rounding.
# north_star's bar is 1e-3 relative.
RTOL_REF, ATOL_REF = 1e-5, 2e-6


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cvpr2020_manet_amd import ops as o
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def chw_view(chw):
    """[C,h,w] device tensor -> the reference caller's permute(1,2,0) view (C-major storage)"""
    return dev(chw).permute(1, 2, 0)


@pytest.mark.parametrize("name", ["global_k1_tm1", "global_k1_tm0", "global_c100_tm1", "global_c100_tm0"])
def test_golden_reference_vectors(ops, oracle, name):
    g = load_golden(name)
    n_ids = int(g["gt_ids"]) + 1
    out = ops.global_match(chw_view(g["ref_chw"]), chw_view(g["qry_chw"]), dev(g["labels"]), n_ids)
    got = out.cpu().numpy().reshape(g["out"].shape)
    np.testing.assert_allclose(got, g["out"], rtol=RTOL_REF, atol=ATOL_REF)
    want = oracle.global_match(np.transpose(g["ref_chw"], (1, 2, 0)), np.transpose(g["qry_chw"], (1, 2, 0)),
                               g["labels"], 1, n_ids=n_ids, test_mode=name.endswith("tm1"))
    np.testing.assert_array_equal(got, want)  # bit-exact vs the oracle


@pytest.mark.parametrize("tm", [1, 0])
def test_golden_stacked_bank_row_major(ops, tm):
    g = load_golden("global_bank2_tm%d" % tm)
    out = ops.global_match(dev(g["bank_hwc"]), chw_view(g["qry_chw"]), dev(g["labels"]), int(g["gt_ids"]) + 1)
    np.testing.assert_allclose(out.cpu().numpy().reshape(g["out"].shape), g["out"], rtol=RTOL_REF, atol=ATOL_REF)


def test_golden_fused_normalize_and_merge(ops):
    g = load_golden("global_c100_tm1")
    n_ids = int(g["gt_ids"]) + 1
    mem = dev(g["mem"].reshape(-1, n_ids))
    out = ops.global_match(chw_view(g["ref_chw"]), chw_view(g["qry_chw"]), dev(g["labels"]), n_ids,
                           normalize=True, mem=mem)
    np.testing.assert_allclose(out.cpu().numpy().reshape(g["merged"].shape), g["merged"], rtol=1e-5, atol=1e-6)
    assert torch.equal(out, mem)  # IntVOS.py:622: the merged map is stored back
    out2 = ops.global_match(chw_view(g["ref_chw"]), chw_view(g["qry_chw"]), dev(g["labels"]), n_ids,
                            normalize=True)
    np.testing.assert_allclose(out2.cpu().numpy().reshape(g["norm"].shape), g["norm"], rtol=1e-5, atol=1e-6)


def _case(seed, h, w, hr, wr, C, n_ids, unlabeled_frac=0.3, scale=0.1):
    rng = np.random.default_rng(seed)
    q = (np.maximum(rng.standard_normal((C, h, w)), 0) * scale).astype(np.float32)
    k = (np.maximum(rng.standard_normal((C, hr, wr)), 0) * scale).astype(np.float32)
    lab = rng.integers(0, n_ids, size=(hr, wr, 1)).astype(np.int32)
    lab[rng.random((hr, wr, 1)) < unlabeled_frac] = -1
    return q, k, lab


@pytest.mark.parametrize("shape", [
    (1, 7, 9, 7, 9, 100, 2),        # smaller than one tile
    (2, 30, 53, 30, 53, 100, 2),    # several query tiles, ragged tails
    (3, 24, 31, 48, 31, 100, 4),    # stacked bank, 4 ids
    (4, 16, 16, 16, 16, 128, 3),    # widest supported C
    (5, 20, 20, 20, 20, 33, 5),     # odd C
    (6, 12, 10, 12, 10, 3, 1),      # tiny C, single id
    (7, 40, 64, 80, 64, 100, 7),    # bigger; many splits per object
])
def test_bit_exact_vs_oracle(ops, oracle, shape):
    seed, h, w, hr, wr, C, n_ids = shape
    q, k, lab = _case(seed, h, w, hr, wr, C, n_ids)
    out = ops.global_match(chw_view(k), chw_view(q), dev(lab), n_ids).cpu().numpy()
    want = oracle.global_match(np.transpose(k, (1, 2, 0)), np.transpose(q, (1, 2, 0)), lab, 1, n_ids=n_ids,
                               test_mode=True).reshape(-1, n_ids)
    np.testing.assert_array_equal(out, want)


def test_edge_cases(ops, oracle):
    q, k, lab = _case(11, 9, 11, 9, 11, 100, 3)
    # (a) nothing labelled: every object at the padding distance (IntVOS.py:81-83) -> normalised 1.0
    none = np.full_like(lab, -1)
    out = ops.global_match(chw_view(k), chw_view(q), dev(none), 3).cpu().numpy()
    assert np.all(out == np.float32(1e20))
    out = ops.global_match(chw_view(k), chw_view(q), dev(none), 3, normalize=True).cpu().numpy()
    assert np.all(out == 1.0)
    # (b) one labelled pixel only
    one = none.copy(); one[4, 5, 0] = 2
    out = ops.global_match(chw_view(k), chw_view(q), dev(one), 3).cpu().numpy()
    want = oracle.global_match(np.transpose(k, (1, 2, 0)), np.transpose(q, (1, 2, 0)), one, 1, n_ids=3).reshape(-1, 3)
    np.testing.assert_array_equal(out, want)
    # (c) labels beyond n_ids are ignored like -1 (masked for every object, IntVOS.py:137)
    big = lab.copy(); big[lab == 2] = 9
    out = ops.global_match(chw_view(k), chw_view(q), dev(big), 2).cpu().numpy()
    want = oracle.global_match(np.transpose(k, (1, 2, 0)), np.transpose(q, (1, 2, 0)), big, 1, n_ids=2).reshape(-1, 2)
    np.testing.assert_array_equal(out, want)
    # (d) query == bank pixel: distance ~0 and possibly slightly negative -- must not be clamped
    out = ops.global_match(chw_view(k), chw_view(k), dev(np.zeros_like(lab)), 1).cpu().numpy()
    want = oracle.global_match(np.transpose(k, (1, 2, 0)), np.transpose(k, (1, 2, 0)), np.zeros_like(lab), 1, n_ids=1).reshape(-1, 1)
    np.testing.assert_array_equal(out, want)
    assert np.abs(out).max() < 1e-5
    # (e) empty bank
    out = ops.global_match(torch.empty(0, 100, device="cuda"), chw_view(q), torch.empty(0, dtype=torch.int32, device="cuda"), 2)
    assert np.all(out.cpu().numpy() == np.float32(1e20))


@pytest.mark.parametrize("tm", [1, 0])
def test_topk_golden_reference_vectors(ops, tm):
    """k_nearest_neighbors = 3 (IntVOS.py:87-94).  TEST_MODE only changes which rows exist."""
    g = load_golden("global_k3_tm%d" % tm)
    n_ids = int(g["gt_ids"]) + 1
    ref, lab = chw_view(g["ref_chw"]).reshape(-1, 16), dev(g["labels"]).reshape(-1)
    if tm:  # the drop-in module filters unlabelled rows first (IntVOS.py:135-136)
        keep = lab != -1
        ref, lab = ref[keep], lab[keep]
    out = ops.global_match(ref, chw_view(g["qry_chw"]), lab, n_ids, k_nearest_neighbors=3)
    np.testing.assert_allclose(out.cpu().numpy().reshape(g["out"].shape), g["out"], rtol=RTOL_REF, atol=ATOL_REF)


@pytest.mark.parametrize("k", [2, 5, 8])
def test_topk_vs_oracle(ops, oracle, k):
    q, kk, lab = _case(31 + k, 30, 41, 60, 41, 100, 4, unlabeled_frac=0.2)
    lab[lab == 3] = -1
    lab[0, 0:3, 0] = 3  # object 3 has only 3 rows: fewer than k for k = 5, 8 -> pad rule
    out = ops.global_match(chw_view(kk), chw_view(q), dev(lab), 5, k_nearest_neighbors=k).cpu().numpy()
    want = oracle.global_match(np.transpose(kk, (1, 2, 0)), np.transpose(q, (1, 2, 0)), lab, k, n_ids=5,
                               test_mode=False).reshape(-1, 5)
    assert np.all(want[:, 4] == 0.0)  # absent object: all invalid -> pad 0 -> mean 0
    np.testing.assert_allclose(out, want, rtol=1e-6, atol=1e-7)  # same distances, mean in the same order
    with pytest.raises(RuntimeError, match="out of range"):
        ops.global_match(chw_view(kk).reshape(-1, 100)[:k - 1], chw_view(q), dev(lab).reshape(-1)[:k - 1], 5,
                         k_nearest_neighbors=k)


# ---- bf16 modes (BASELINE cfg3 / cfg5 arithmetic) -----------------------------------------------------
# MANET_COMPUTE_BF16: the reference formula evaluated on embeddings rounded to bf16 (norms included),
# products exact in fp32, fp32 accumulation inside v_mfma_f32_32x32x16_bf16 in an order that is not a
# plain chain -> compared with the oracle's quant_bf16 mode to accumulation-order rounding.
BF16_RTOL, BF16_ATOL = 1e-5, 3e-6
# MANET_COMPUTE_BF16X3 (hi/lo split, 3 MFMAs): must meet north_star's fp32 bar against the UNQUANTISED
# oracle: 1e-3 relative.  Measured worst case is ~1e-5 of |q|^2+|k|^2, asserted 10x tighter than the bar.
X3_RTOL, X3_ATOL = 1e-4, 3e-6


@pytest.mark.parametrize("shape", [
    (41, 7, 9, 7, 9, 100, 2),
    (42, 30, 53, 30, 53, 100, 4),     # cfg3-like: 4 ids
    (43, 40, 64, 80, 64, 100, 6),     # cfg5-like: 6 ids, stacked bank
    (44, 16, 16, 16, 16, 128, 3),
    (45, 12, 10, 12, 10, 20, 2),
])
def test_bf16_vs_quantised_oracle(ops, oracle, shape):
    seed, h, w, hr, wr, C, n_ids = shape
    q, k, lab = _case(seed, h, w, hr, wr, C, n_ids)
    out = ops.global_match(chw_view(k), chw_view(q), dev(lab), n_ids, compute="bf16").cpu().numpy()
    want = oracle.global_match(np.transpose(k, (1, 2, 0)), np.transpose(q, (1, 2, 0)), lab, 1, n_ids=n_ids,
                               quant_bf16=True).reshape(-1, n_ids)
    np.testing.assert_allclose(out, want, rtol=BF16_RTOL, atol=BF16_ATOL)
    # and it is the plain fp32 result of the rounded embeddings (bit-exact kernel vs itself is not
    # implied; same tolerance)
    qr, kr = oracle.bf16_round(q), oracle.bf16_round(k)
    out32 = ops.global_match(chw_view(kr), chw_view(qr), dev(lab), n_ids, compute="f32").cpu().numpy()
    np.testing.assert_allclose(out, out32, rtol=BF16_RTOL, atol=BF16_ATOL)


@pytest.mark.parametrize("shape", [
    (51, 7, 9, 7, 9, 100, 2),
    (52, 30, 53, 30, 53, 100, 4),
    (53, 40, 64, 80, 64, 100, 6),
    (54, 16, 16, 16, 16, 128, 3),
])
def test_bf16x3_meets_fp32_bar(ops, oracle, shape):
    seed, h, w, hr, wr, C, n_ids = shape
    q, k, lab = _case(seed, h, w, hr, wr, C, n_ids)
    out = ops.global_match(chw_view(k), chw_view(q), dev(lab), n_ids, compute="bf16x3").cpu().numpy()
    want = oracle.global_match(np.transpose(k, (1, 2, 0)), np.transpose(q, (1, 2, 0)), lab, 1,
                               n_ids=n_ids).reshape(-1, n_ids)
    np.testing.assert_allclose(out, want, rtol=X3_RTOL, atol=X3_ATOL)


def test_bf16_golden_reference_vectors(ops):
    """against the reference's own fp32 outputs: bf16x3 inside the fp32 bar; plain bf16 deviates by the
    input rounding (2^-9 relative per embedding value) -- bounded here, documented in DESIGN.md"""
    g = load_golden("global_c100_tm1")
    n_ids = int(g["gt_ids"]) + 1
    x3 = ops.global_match(chw_view(g["ref_chw"]), chw_view(g["qry_chw"]), dev(g["labels"]), n_ids, compute="bf16x3")
    np.testing.assert_allclose(x3.cpu().numpy().reshape(g["out"].shape), g["out"], rtol=1e-3, atol=3e-6)
    bf = ops.global_match(chw_view(g["ref_chw"]), chw_view(g["qry_chw"]), dev(g["labels"]), n_ids, compute="bf16",
                          normalize=True)
    assert np.abs(bf.cpu().numpy().reshape(g["norm"].shape) - g["norm"]).max() < 2e-3  # on the normalised map


def test_full_size_properties_cfg5_bf16(ops):
    """BASELINE cfg5: 720p grid 180x320, C=100, 10-frame bank (M=576000), 6 ids, bf16."""
    torch.manual_seed(20200614 + 5)
    C, h, w, T, n_ids = 100, 180, 320, 10, 6
    q = torch.relu(torch.randn(C, h, w, device="cuda")) * 0.1
    bank = torch.relu(torch.randn(T, C, h, w, device="cuda")) * 0.1
    bank[7] = q
    N = h * w
    lab = torch.randint(0, n_ids, (T * N,), device="cuda", dtype=torch.int32)
    bank_rows = bank.permute(0, 2, 3, 1).reshape(-1, C)
    for mode, tol in (("bf16", 2e-6), ("bf16x3", 2e-5)):
        out = ops.global_match(bank_rows, q.permute(1, 2, 0), lab, n_ids, compute=mode)
        own = lab[7 * N:8 * N].long()
        assert out.gather(1, own[:, None]).abs().max().item() < tol  # self match ~ 0
        perm = torch.randperm(T * N, device="cuda")
        outp = ops.global_match(bank_rows[perm], q.permute(1, 2, 0), lab[perm], n_ids, compute=mode)
        assert torch.equal(out, outp)  # row order never matters, bit for bit
    # fp64 spot check of bf16 on the rounded embeddings
    out = ops.global_match(bank_rows, q.permute(1, 2, 0), lab, n_ids, compute="bf16")
    idx = torch.randint(0, N, (32,), device="cuda")
    qs = q.permute(1, 2, 0).reshape(-1, C)[idx].bfloat16().double()
    kb = bank_rows.bfloat16().double()
    d = (qs * qs).sum(1, keepdim=True) + (kb * kb).sum(1)[None] - 2 * qs @ kb.t()
    for o in range(n_ids):
        dm = d.masked_fill((lab != o)[None], float("inf")).min(dim=1).values
        assert torch.allclose(out[idx, o].double(), dm, rtol=1e-3, atol=3e-6)


def test_prepared_bank_equals_one_shot(ops):
    q, k, lab = _case(21, 30, 40, 60, 40, 100, 3)
    one = ops.global_match(chw_view(k), chw_view(q), dev(lab), 3)
    bank = ops.PreparedBank(chw_view(k), dev(lab), 3)
    two = bank.match(chw_view(q))
    three = bank.match(chw_view(q))  # reusable
    assert torch.equal(one, two) and torch.equal(two, three)


def test_errors_are_loud(ops):
    q, k, lab = _case(5, 8, 8, 8, 8, 16, 2)
    with pytest.raises(RuntimeError):
        ops.global_match(torch.from_numpy(k).permute(1, 2, 0), chw_view(q), dev(lab), 2)  # CPU tensor
    with pytest.raises(RuntimeError):
        ops.global_match(chw_view(k), chw_view(q), dev(lab), 100)  # n_ids out of range
    with pytest.raises(RuntimeError, match="k_nn > 1 needs"):
        ops.global_match(chw_view(k), chw_view(q), dev(lab), 2, k_nearest_neighbors=2, compute="bf16")
    with pytest.raises(ValueError):
        ops.global_match(chw_view(k), chw_view(q), dev(lab[:4]), 2)


def test_full_size_properties_cfg2(ops):
    """BASELINE cfg2: 480p grid 120x214, C=100, 5-frame fully-labelled bank (M=128400), 2 ids.
    The oracle cannot run this in seconds; check properties that pin the result instead."""
    torch.manual_seed(20200614 + 2)
    C, h, w, T, n_ids = 100, 120, 214, 5, 2
    q = torch.relu(torch.randn(C, h, w, device="cuda")) * 0.1
    bank = torch.relu(torch.randn(T, C, h, w, device="cuda")) * 0.1
    bank[2] = q  # frame 2 of the bank IS the query frame
    lab = torch.randint(0, n_ids, (T * h * w,), device="cuda", dtype=torch.int32)
    bank_rows = bank.permute(0, 2, 3, 1).reshape(-1, C)  # row-major stacked bank
    out = ops.global_match(bank_rows, q.permute(1, 2, 0), lab, n_ids)
    N = h * w
    # (1) every query pixel is in the bank with label L -> its min distance for L is ~0
    own = lab[2 * N:3 * N].long()
    d_own = out.gather(1, own[:, None])[:, 0]
    assert d_own.abs().max().item() < 1e-5
    # (2) min over objects == unmasked min (single id covering everything)
    allz = torch.zeros_like(lab)
    out1 = ops.global_match(bank_rows, q.permute(1, 2, 0), allz, 1)
    assert torch.equal(out.min(dim=1).values, out1[:, 0])
    # (3) permuting the bank rows (and labels) does not change a single bit
    perm = torch.randperm(T * N, device="cuda")
    outp = ops.global_match(bank_rows[perm], q.permute(1, 2, 0), lab[perm], n_ids)
    assert torch.equal(out, outp)
    # (4) a random sample of query pixels against a torch fp64 evaluation of the reference formula
    idx = torch.randint(0, N, (64,), device="cuda")
    qs = q.permute(1, 2, 0).reshape(-1, C)[idx].double()
    d = (qs * qs).sum(1, keepdim=True) + (bank_rows.double() ** 2).sum(1)[None] - 2 * qs @ bank_rows.double().t()
    for o in range(n_ids):
        dm = d.masked_fill((lab != o)[None], float("inf")).min(dim=1).values
        # absolute: |d| ~ 0.1 and fp32 cancellation noise ~1e-7; relative bar of north_star is 1e-3
        assert torch.allclose(out[idx, o].double(), dm, rtol=1e-3, atol=2e-6)


@pytest.mark.parametrize("mode", ["f32", "bf16", "bf16x3"])
def test_run_to_run_determinism(ops, mode):
    """atomicMin across bank splits and the stable bank sort make every mode bit-reproducible"""
    q, k, lab = _case(61, 60, 80, 120, 80, 100, 3)
    outs = [ops.global_match(chw_view(k), chw_view(q), dev(lab), 3, compute=mode) for _ in range(4)]
    assert all(torch.equal(outs[0], o) for o in outs[1:])


def test_hip_graph_capture_of_a_propagated_frame(ops):
    """the C ABI never allocates or synchronises, so a frame's launch sequence can be captured in a HIP
    graph by the caller and replayed (same results as eager)"""
    q, k, lab = _case(62, 40, 50, 80, 50, 100, 3)
    qv, kv, lv = chw_view(q), chw_view(k), dev(lab)
    prev, plab = chw_view(k[:, :40, :]), dev(lab[:40])
    mem = torch.ones(40 * 50, 3, device="cuda")
    eager_g = ops.global_match(kv, qv, lv, 3, normalize=True, mem=mem.clone())
    eager_l = ops.local_match(prev, qv, plab, 3, 4)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):  # warm the per-stream workspaces outside the capture
        ops.global_match(kv, qv, lv, 3, normalize=True, mem=mem.clone())
        ops.local_match(prev, qv, plab, 3, 4)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    gmem = mem.clone()
    with torch.cuda.graph(graph, stream=s):
        g = ops.global_match(kv, qv, lv, 3, normalize=True, mem=gmem)
        l = ops.local_match(prev, qv, plab, 3, 4)
    for _ in range(3):
        gmem.fill_(1.0)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(g, eager_g) and torch.equal(l, eager_l)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [300, 6000, 17035, 25680])
def test_device_side_split_decision_gives_the_same_bits(rows, monkeypatch):
    """r5: for small / mid-size banks the fp32 kernel swaps the host's many short splits (sized from the bank's upper-bound tile
    count) for ONE round of long ones (split_of_block); a minimum is order-independent: the results are the host-split run's bits"""
    monkeypatch.setenv("MANET_TUNING", "1")
    import torch
    from cvpr2020_manet_amd import _lib, ops
    lib = _lib.load()
    torch.manual_seed(rows)
    C, H, W = 100, 120, 214
    q = torch.relu(torch.randn(H, W, C, device="cuda")) * 0.1
    bank = torch.relu(torch.randn(H * W, C, device="cuda")) * 0.1
    lab = torch.full((H * W,), -1, dtype=torch.int32, device="cuda")
    idx = torch.randperm(H * W, device="cuda")[:rows]
    lab[idx] = torch.randint(0, 3, (rows,), device="cuda", dtype=torch.int32)
    pb = ops.PreparedBank(bank, lab, 3)
    try:
        assert lib.manet_tune_set(10, 1) == 0
        host = pb.match(q, normalize=True)
    finally:
        lib.manet_tune_set(10, -2 ** 31)
    assert torch.equal(pb.match(q, normalize=True), host)
