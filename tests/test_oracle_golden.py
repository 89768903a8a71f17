"""Pins the CPU oracle (oracle/manet_oracle.c) against golden vectors produced by running the
reference itself (oracle/gen_golden.py -> tests/golden/*.npz).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden

# The reference's torch.matmul / torch.sum accumulate in an unspecified order, the oracle in a
# k-ascending fmaf chain: agreement is to fp32 rounding.  north_star's bar is 1e-3 relative.
RTOL = 1e-5
ATOL = 2e-6


def hwc(chw):
    """the reference's callers pass permute(1,2,0) views of C-major storage"""
    return np.transpose(chw, (1, 2, 0))


@pytest.mark.parametrize("tm", [1, 0])
@pytest.mark.parametrize("case", ["global_k1", "global_k3"])
def test_global_match(oracle, case, tm):
    g = load_golden("%s_tm%d" % (case, tm))
    out = oracle.global_match(hwc(g["ref_chw"]), hwc(g["qry_chw"]), g["labels"], int(g["k"]),
                              n_ids=int(g["gt_ids"]) + 1, test_mode=bool(tm))
    assert out.shape == g["out"].shape
    # the absent object id must come out as exactly the padding distance (IntVOS.py:81-83)
    # (top-k path: every candidate is invalid -> pad distance max(0)=0 -> mean 0, IntVOS.py:89-94)
    absent = np.float32(1e20) if int(g["k"]) == 1 else np.float32(0.0)
    assert np.all(out[..., 3, 0] == absent) and np.all(g["out"][..., 3, 0] == absent)
    np.testing.assert_allclose(out, g["out"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("tm", [1, 0])
def test_global_match_ids_from_labels(oracle, tm):
    g = load_golden("global_k1_noids_tm%d" % tm)
    out = oracle.global_match(hwc(g["ref_chw"]), hwc(g["qry_chw"]), g["labels"], 1, n_ids=None,
                              test_mode=bool(tm))
    assert list(g["ids"]) == list(range(out.shape[3]))
    np.testing.assert_allclose(out, g["out"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("tm", [1, 0])
def test_global_match_stacked_bank(oracle, tm):
    g = load_golden("global_bank2_tm%d" % tm)
    out = oracle.global_match(g["bank_hwc"], hwc(g["qry_chw"]), g["labels"], 1,
                              n_ids=int(g["gt_ids"]) + 1, test_mode=bool(tm))
    np.testing.assert_allclose(out, g["out"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("tm", [1, 0])
def test_global_match_c100_and_aggregation(oracle, tm):
    g = load_golden("global_c100_tm%d" % tm)
    out = oracle.global_match(hwc(g["ref_chw"]), hwc(g["qry_chw"]), g["labels"], 1,
                              n_ids=int(g["gt_ids"]) + 1, test_mode=bool(tm))
    np.testing.assert_allclose(out, g["out"], rtol=RTOL, atol=ATOL)
    norm, _ = oracle.normalize_merge(g["out"], None, normalize=True)
    np.testing.assert_allclose(norm, g["norm"], rtol=1e-5, atol=1e-6)
    merged, mem = oracle.normalize_merge(g["out"], g["mem"], normalize=True)
    np.testing.assert_allclose(merged, g["merged"], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(merged, mem)  # IntVOS.py:622 stores the merged map


LOCAL_CASES = ["C16_12x15_d2", "C16_12x15_d4", "C100_10x14_d3", "C8_9x11_d1"]


@pytest.mark.parametrize("ds", [1, 0])
@pytest.mark.parametrize("case", LOCAL_CASES)
def test_local_dist_and_match(oracle, case, ds):
    g = load_golden("local_ds%d_%s" % (ds, case))
    d, n_ids = int(g["d"]), int(g["n_ids"])
    dist = oracle.local_dist(hwc(g["cur_chw"]), hwc(g["prev_chw"]), d, downsample=bool(ds))
    assert dist.shape == g["dist"].shape
    ref = g["dist"]
    # outside the image: inf (raw, ds0) / exactly 1.0 (normalised, ds1)
    assert np.array_equal(np.isinf(dist), np.isinf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(dist[fin], ref[fin], rtol=1e-5, atol=2e-6)
    out = oracle.local_masked_min(ref, g["labels"], d, n_ids)
    np.testing.assert_array_equal(out, g["out"])  # pure selection on the reference's own volume
    out2 = oracle.local_match(hwc(g["prev_chw"]), hwc(g["cur_chw"]), g["labels"], n_ids, d,
                              downsample=bool(ds))
    np.testing.assert_allclose(out2, g["out"], rtol=1e-5, atol=2e-6)


def test_correlation_tied_to_reference_distance(oracle):
    """correlation_package cannot be built here (CUDA only).  For pad=d, K=1, max_disp=d,
    s1=s2=1:  xs + ys_shift - 2*C*corr must equal the reference's a9 distances inside the image
    (SURVEY.md 8c), which ties the op contract to an importable reference function."""
    g = load_golden("correlation_tie")
    a, b, d = g["in1"], g["in2"], int(g["d"])
    C, h, w = a.shape
    corr = oracle.correlation_forward(a[None], b[None], d, 1, d, 1, 1)  # [1,P*P,h,w]
    assert corr.shape == (1, (2 * d + 1) ** 2, h, w)
    xs = (a.astype(np.float64) ** 2).sum(0)
    ys = (b.astype(np.float64) ** 2).sum(0)
    ref = g["dist"]
    P = 2 * d + 1
    for dy in range(P):
        for dx in range(P):
            for y in range(h):
                for x in range(w):
                    yy, xx = y + dy - d, x + dx - d
                    r = ref[y, x, dy * P + dx]
                    c = corr[0, dy * P + dx, y, x] * C
                    if 0 <= yy < h and 0 <= xx < w:
                        assert abs(xs[y, x] + ys[yy, xx] - 2 * c - r) < 1e-4 * max(1.0, r)
                    else:
                        assert np.isinf(r) and c == 0.0  # zero padding -> zero correlation


def test_correlation_shape_rule(oracle):
    # correlation_cuda.cc:25-34
    assert oracle.correlation_out_dims(20, 30, 4, 1, 4, 1, 1) == (81, 20, 30)
    assert oracle.correlation_out_dims(20, 30, 20, 1, 20, 1, 2) == (441, 20, 30)
    assert oracle.correlation_out_dims(21, 31, 3, 3, 3, 2, 1) == (49, 10, 15)


def test_correlation_kernel_window_bruteforce(oracle):
    rng = np.random.default_rng(3)
    B, C, H, W = 2, 5, 9, 8
    a = rng.standard_normal((B, C, H, W)).astype(np.float32)
    b = rng.standard_normal((B, C, H, W)).astype(np.float32)
    pad, K, md, s1, s2 = 3, 3, 2, 2, 2
    out = oracle.correlation_forward(a, b, pad, K, md, s1, s2)
    ap = np.zeros((B, C, H + 2 * pad, W + 2 * pad)); ap[:, :, pad:pad + H, pad:pad + W] = a
    bp = np.zeros_like(ap); bp[:, :, pad:pad + H, pad:pad + W] = b
    r, kr = md // s2, (K - 1) // 2
    oc, oh, ow = out.shape[1:]
    for n in range(B):
        for oy in range(oh):
            for ox in range(ow):
                y1, x1 = oy * s1 + md, ox * s1 + md
                for tj in range(-r, r + 1):
                    for ti in range(-r, r + 1):
                        y2, x2 = y1 + tj * s2, x1 + ti * s2
                        acc = 0.0
                        for j in range(-kr, kr + 1):
                            for i in range(-kr, kr + 1):
                                acc += (ap[n, :, y1 + j, x1 + i] * bp[n, :, y2 + j, x2 + i]).sum()
                        tc = (tj + r) * (2 * r + 1) + (ti + r)
                        assert abs(out[n, tc, oy, ox] - acc / (K * K * C)) < 1e-5
