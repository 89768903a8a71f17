"""The block -> (query tile, bank split) maps of the global-match kernels (split_of_block, csrc/global_match.hip): every map
visits every (tile, split) exactly once, and a minimum does not care in which order its partial results arrive -- so every
map must give the shipped (automatic) map's bits, for the fp32 and the bf16 kernel, on banks whose split count is and is
not a multiple of the splits-fastest group."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
INT32_MIN = -2 ** 31


@pytest.mark.parametrize("compute", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(6000, 60000, 3), (9000, 150000, 2), (2500, 40000, 5)])
def test_every_block_map_gives_the_same_bits(compute, shape):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cvpr2020_manet_amd import _lib, ops
    N, M, n_ids = shape
    g = torch.Generator(device="cuda").manual_seed(N + M)
    q = torch.relu(torch.randn(N, 100, generator=g, device="cuda")) * 0.2
    k = torch.relu(torch.randn(M, 100, generator=g, device="cuda")) * 0.2
    lab = torch.randint(-1, n_ids, (M,), generator=g, device="cuda", dtype=torch.int32)
    lib = _lib.load()
    os.environ["MANET_TUNING"] = "1"  # the setters refuse without the opt-in
    want = ops.global_match(k, q, lab, n_ids, compute=compute)  # the automatic map
    try:
        for bm in (0, 1, 2, 4, 5, 6, 7):
            _lib.check(lib.manet_tune_set(0, bm), "manet_tune_set")
            assert torch.equal(ops.global_match(k, q, lab, n_ids, compute=compute), want), bm
        for splits in (8, 24, 40):  # forced split counts: S / 8 = 1, 3, 5 against groups of 2..5
            _lib.check(lib.manet_tune_set(1, splits), "manet_tune_set")
            for bm in (4, 5, 7):
                _lib.check(lib.manet_tune_set(0, bm), "manet_tune_set")
                assert torch.equal(ops.global_match(k, q, lab, n_ids, compute=compute), want), (splits, bm)
    finally:
        _lib.check(lib.manet_tune_set(0, INT32_MIN), "manet_tune_set")  # back to "not set": the automatic map
        _lib.check(lib.manet_tune_set(1, INT32_MIN), "manet_tune_set")
    assert torch.equal(ops.global_match(k, q, lab, n_ids, compute=compute), want)
