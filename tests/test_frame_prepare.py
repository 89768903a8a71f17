"""manet_frame_prepare (the producer side): one read of a frame's embedding writes the query operand
image of the global match AND the padded pooled plane + tile table of the local match.  The prepared path must be
bit-identical to the per-call path (which is pinned to the oracle / the reference's golden vectors elsewhere):
  PreparedBank.match(PreparedFrame)            == PreparedBank.match(embedding)        (every arithmetic mode)
  local_match_frames(prev_frame, cur_frame)    == local_match(prev, cur)               (every window radius class)
for fp32 and 2-byte storage, odd grids (pooling floors, IntVOS.py:282-284), strided batches, and against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 2e-6


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cvpr2020_manet_amd import ops as o
    return o


def _frames(B, C, h, w, seed, dtype=torch.float32, scale=0.2):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.relu(torch.randn(B, C, h, w, generator=g, device="cuda")) * scale).to(dtype)


@pytest.mark.parametrize("compute", ["f32", "bf16", "bf16x3"])
@pytest.mark.parametrize("shape", [(100, 60, 107), (16, 9, 11), (100, 24, 31), (37, 2, 2), (128, 7, 70)])
def test_global_match_on_prepared_frame_is_bit_identical(ops, compute, shape):
    C, h, w = shape
    n_ids = 3
    emb = _frames(2, C, h, w, 11 + h)
    g = torch.Generator(device="cuda").manual_seed(5)
    bank = torch.relu(torch.randn(700, C, generator=g, device="cuda")) * 0.2
    lab = torch.randint(-1, n_ids, (700,), generator=g, device="cuda", dtype=torch.int32)
    pb = ops.PreparedBank(bank, lab, n_ids, compute=compute)
    for storage in (torch.float32, torch.bfloat16):
        e = emb.to(storage)
        frames = ops.prepare_frames(e, compute=compute, max_distance=-1)
        for i in range(2):
            want = pb.match(e[i].permute(1, 2, 0))
            got = pb.match(frames[i])
            assert torch.equal(got, want)
            # twice on the same armed workspace, with the fused epilogue: the keys were re-armed by the first call
            mem1, mem2 = torch.full((h * w, n_ids), 0.3, device="cuda"), torch.full((h * w, n_ids), 0.3, device="cuda")
            want_n = pb.match(e[i].permute(1, 2, 0), normalize=True, mem=mem1)
            got_n = pb.match(frames[i], normalize=True, mem=mem2)
            assert torch.equal(got_n, want_n) and torch.equal(mem1, mem2)


def test_armed_workspace_survives_changes_of_shape_and_id_count(ops):
    C = 100
    g = torch.Generator(device="cuda").manual_seed(9)
    for (h, w, n_ids) in ((20, 30, 2), (40, 50, 5), (10, 12, 1), (40, 50, 3)):
        e = _frames(1, C, h, w, h)[0]
        bank = torch.relu(torch.randn(300, C, generator=g, device="cuda")) * 0.2
        lab = torch.randint(0, n_ids, (300,), generator=g, device="cuda", dtype=torch.int32)
        pb = ops.PreparedBank(bank, lab, n_ids)
        f = ops.prepare_frames(e)
        assert torch.equal(pb.match(f), pb.match(e.permute(1, 2, 0)))


@pytest.mark.parametrize("d", [0, 1, 2, 4, 6, 7, 9, 10, 11, 12])
def test_local_match_on_prepared_frames_is_bit_identical(ops, oracle, d):
    for (C, h, w, n_ids, storage) in ((100, 60, 107, 3, torch.float32), (58, 53, 71, 11, torch.float32),
                                      (100, 24, 30, 2, torch.bfloat16), (7, 9, 100, 2, torch.float32)):
        e = _frames(2, C, h, w, 100 * d + h, storage)
        g = torch.Generator(device="cuda").manual_seed(d)
        lab = torch.randint(-1, n_ids + 1, (h, w), generator=g, device="cuda", dtype=torch.int32)
        prev, cur = e[0].permute(1, 2, 0), e[1].permute(1, 2, 0)
        want = ops.local_match(prev, cur, lab, n_ids, d)
        fr = ops.prepare_frames(e, compute="f32", max_distance=d)
        got = ops.local_match_frames(fr[0], fr[1], lab, n_ids)
        assert torch.equal(got, want)
        # a frame against itself (int_seghead, IntVOS.py:709-711) and the pre-set `out` form
        out = torch.empty(h, w, n_ids, device="cuda")
        fr2 = ops.prepare_frames(e[1], compute="f32", max_distance=d, preset=out, preset_value=1.0)
        got_self = ops.local_match_frames(fr2, fr2, lab, n_ids, out=out, out_is_preset=True)
        assert torch.equal(got_self, ops.local_match(cur, cur, lab, n_ids, d))
    # and against the oracle itself on the last case
    ref = oracle.local_match(prev.float().cpu().numpy(), cur.float().cpu().numpy(), lab.cpu().numpy().reshape(h, w, 1),
                             n_ids, d, downsample=True).reshape(h, w, n_ids)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=RTOL, atol=ATOL)


def test_strided_batch_and_both_operands_from_one_prepare(ops):
    """extract_feature's batch as a non-contiguous view (channel-sliced, batch-strided); the one prepared object serves
    the global AND the local match of a propagation step"""
    big = _frames(5, 120, 30, 44, 77)
    e = big[::2, 10:110]  # [3, 100, 30, 44], strides (2*C*h*w, h*w, w, 1) with an offset
    n_ids, d = 4, 4
    g = torch.Generator(device="cuda").manual_seed(3)
    lab = torch.randint(0, n_ids, (30, 44), generator=g, device="cuda", dtype=torch.int32)
    bank_lab = torch.randint(0, n_ids, (30 * 44,), generator=g, device="cuda", dtype=torch.int32)
    for compute in ("f32", "bf16"):
        fr = ops.prepare_frames(e, compute=compute, max_distance=d)
        pb = ops.PreparedBank(e[0].permute(1, 2, 0), bank_lab, n_ids, compute=compute)
        for i in (1, 2):
            assert torch.equal(pb.match(fr[i]), pb.match(e[i].permute(1, 2, 0)))
            assert torch.equal(ops.local_match_frames(fr[i - 1], fr[i], lab, n_ids),
                               ops.local_match(e[i - 1].permute(1, 2, 0), e[i].permute(1, 2, 0), lab, n_ids, d))


def test_full_size_cfg2_and_cfg5(ops):
    for (h, w, d, n_ids, storage) in ((120, 214, 12, 2, torch.float32), (180, 320, 4, 6, torch.bfloat16)):
        e = _frames(2, 100, h, w, h, storage, scale=0.1)
        lab = torch.randint(0, n_ids, (h, w), device="cuda", dtype=torch.int32)
        fr = ops.prepare_frames(e, compute="bf16" if storage == torch.bfloat16 else "f32", max_distance=d)
        assert torch.equal(ops.local_match_frames(fr[0], fr[1], lab, n_ids),
                           ops.local_match(e[0].permute(1, 2, 0), e[1].permute(1, 2, 0), lab, n_ids, d))


def test_errors_are_loud(ops):
    e = _frames(2, 100, 20, 30, 1)
    fa = ops.prepare_frames(e[0], max_distance=4)
    fb = ops.prepare_frames(e[1], max_distance=2)
    lab = torch.zeros(20, 30, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        ops.local_match_frames(fa, fb, lab, 2)
    with pytest.raises(ValueError):
        ops.local_match_frames(ops.prepare_frames(e[0]), ops.prepare_frames(e[1]), lab, 2)  # no pooled plane
    with pytest.raises(RuntimeError):
        ops.prepare_frames(e.cpu())
    pb = ops.PreparedBank(e[0].permute(1, 2, 0), lab, 2, compute="bf16")
    with pytest.raises(ValueError):
        pb.match(fa)  # packed for f32 arithmetic
