"""Seeded random sweeps over shapes and label distributions (ragged sizes, empty objects, sparse scribbles,
row-major and C-major sources) -- the HIP path against the oracle.  Global fp32: bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
# MANET_SWEEP_SCALE=20 runs 20x the seeds (a soak; the default keeps the suite short)
SCALE = int(os.environ.get("MANET_SWEEP_SCALE", "1"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("seed", range(12 * SCALE))
def test_global_random_shapes_bit_exact(oracle, seed):
    from cvpr2020_manet_amd import ops
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.choice([1, 7, 16, 33, 64, 100, 101, 104, 128]))
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 60))
    T = int(rng.integers(1, 4))
    hr, wr = T * int(rng.integers(1, 30)), int(rng.integers(1, 50))
    n_ids = int(rng.integers(1, 12))
    q = (rng.standard_normal((C, h, w)) * rng.choice([0.05, 1.0])).astype(np.float32)
    k = (rng.standard_normal((C, hr, wr)) * rng.choice([0.05, 1.0])).astype(np.float32)
    lab = rng.integers(-1, n_ids + 2, size=(hr, wr, 1)).astype(np.int32)  # includes -1 and ids beyond n_ids
    if rng.random() < 0.5:  # sparse scribbles
        lab[rng.random(lab.shape) < 0.9] = -1
    if rng.random() < 0.3 and n_ids > 1:  # an object without any pixel
        lab[lab == n_ids - 1] = -1
    if rng.random() < 0.5:  # C-major views (the reference's callers) ...
        qt, kt = dev(q).permute(1, 2, 0), dev(k).permute(1, 2, 0)
    else:                   # ... or row-major contiguous
        qt, kt = dev(np.transpose(q, (1, 2, 0))), dev(np.transpose(k, (1, 2, 0)))
    got = ops.global_match(kt, qt, dev(lab), n_ids).cpu().numpy()
    want = oracle.global_match(np.transpose(k, (1, 2, 0)), np.transpose(q, (1, 2, 0)), lab, 1, n_ids=n_ids,
                               test_mode=bool(seed & 1)).reshape(-1, n_ids)
    np.testing.assert_array_equal(got, want)
    # the other arithmetic modes stay inside their documented bars on the same case
    x3 = ops.global_match(kt, qt, dev(lab), n_ids, compute="bf16x3").cpu().numpy()
    scale = 1.0 + np.abs(want[want < 1e19]).max() if (want < 1e19).any() else 1.0
    ok = want < 1e19
    assert np.array_equal(x3 >= 1e19, ~ok)
    # (split-bf16: <= 2^-16 relative per PRODUCT, i.e. an absolute bound that scales with |q||k| -- not with the distance, which
    # may be tiny beside the norms: seed 153 of a 25x soak, C = 1)
    qk = float(np.sqrt((q.astype(np.float64) ** 2).sum(0).max() * (k.astype(np.float64) ** 2).sum(0).max()))
    np.testing.assert_allclose(x3[ok], want[ok], rtol=1e-4, atol=2e-5 * scale + 5e-5 * qk)


@pytest.mark.parametrize("seed", range(10 * SCALE))
def test_local_random_shapes(oracle, seed):
    from cvpr2020_manet_amd import ops
    rng = np.random.default_rng(2000 + seed)
    C = int(rng.choice([1, 5, 16, 100]))
    d = int(rng.integers(0, 13))
    ds = bool(rng.integers(0, 2))
    h, w = int(rng.integers(2, 40)), int(rng.integers(2, 50))
    n_ids = int(rng.integers(1, 11))
    prev = (np.maximum(rng.standard_normal((C, h, w)), 0) * 0.2).astype(np.float32)
    cur = (np.maximum(rng.standard_normal((C, h, w)), 0) * 0.2).astype(np.float32)
    lab = rng.integers(-1, n_ids, size=(h, w, 1)).astype(np.int32)
    got = ops.local_match(dev(prev).permute(1, 2, 0), dev(cur).permute(1, 2, 0), dev(lab), n_ids, d,
                          downsample=ds).cpu().numpy()
    want = oracle.local_match(np.transpose(prev, (1, 2, 0)), np.transpose(cur, (1, 2, 0)), lab, n_ids, d,
                              downsample=ds).reshape(h, w, n_ids)
    assert np.array_equal(np.isinf(got), np.isinf(want))
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("seed", range(4 * SCALE))
def test_topk_random(oracle, seed):
    from cvpr2020_manet_amd import ops
    rng = np.random.default_rng(3000 + seed)
    C, h, w, hr, wr = 100, int(rng.integers(3, 30)), int(rng.integers(3, 40)), int(rng.integers(3, 40)), int(rng.integers(3, 40))
    n_ids, kk = int(rng.integers(1, 6)), int(rng.integers(2, 9))
    q = (np.maximum(rng.standard_normal((C, h, w)), 0) * 0.1).astype(np.float32)
    k = (np.maximum(rng.standard_normal((C, hr, wr)), 0) * 0.1).astype(np.float32)
    lab = rng.integers(0, n_ids, size=(hr, wr, 1)).astype(np.int32)
    got = ops.global_match(dev(k).permute(1, 2, 0), dev(q).permute(1, 2, 0), dev(lab), n_ids,
                           k_nearest_neighbors=kk).cpu().numpy()
    want = oracle.global_match(np.transpose(k, (1, 2, 0)), np.transpose(q, (1, 2, 0)), lab, kk, n_ids=n_ids,
                               test_mode=False).reshape(-1, n_ids)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7)
