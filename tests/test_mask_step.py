"""the driver's mask step (bilinear upsample + argmax + nearest resize back),
test.py:253-255 + IntVOS.py:598-599.  Integer outputs: the bar is exact equality."""
import numpy as np
import pytest
import torch

from conftest import load_golden

CASES = ["mask_step_10x13_to_40x52_n3", "mask_step_12x21_to_47x85_n5", "mask_step_30x54_to_120x214_n2"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_calls(oracle, name):
    g = load_golden(name)
    mask, small = oracle.upsample_argmax(g["logits"], g["size"])
    np.testing.assert_array_equal(mask, g["mask"])
    np.testing.assert_array_equal(small, g["small"])


def test_oracle_on_the_end_to_end_vector(oracle):
    """the reference class's own logits -> the mask the reference driver derived from them"""
    g = load_golden("e2e_tiny")
    mask, _ = oracle.upsample_argmax(g["int_logits"], g["int_pred"].shape[1:])
    np.testing.assert_array_equal(mask, g["int_pred"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_matches_reference_calls(name):
    from cvpr2020_manet_amd import ops
    g = load_golden(name)
    mask, small = ops.upsample_argmax(torch.from_numpy(g["logits"]).cuda(), g["size"])
    assert mask.dtype == torch.int64 and small.dtype == torch.int32
    np.testing.assert_array_equal(mask.cpu().numpy(), g["mask"])
    np.testing.assert_array_equal(small.cpu().numpy(), g["small"])


@pytest.mark.gpu
def test_gpu_full_size_vs_torch_ops_and_oracle(oracle):
    """480p: [1,4,120,214] -> (480,854).  Equality with the framework ops the reference calls, except where
    two ids tie to the last bit of a different (but equally valid) fp32 evaluation order."""
    from cvpr2020_manet_amd import ops
    torch.manual_seed(5)
    logits = torch.randn(1, 4, 120, 214, device="cuda")
    mask, small = ops.upsample_argmax(logits, (480, 854))
    up = torch.nn.functional.interpolate(logits, size=(480, 854), mode="bilinear", align_corners=True)
    ref_mask = torch.argmax(up, dim=1)
    diff = mask != ref_mask
    if diff.any():  # only exact-to-rounding ties may differ
        top2 = up[0].topk(2, dim=0).values
        assert ((top2[0] - top2[1])[diff[0]] < 1e-5).all()
    assert diff.float().mean().item() < 1e-4
    ref_small = torch.nn.functional.interpolate(mask.unsqueeze(0).float(), size=(120, 214), mode="nearest").int()
    assert torch.equal(small, ref_small)
    om, osm = oracle.upsample_argmax(logits.cpu().numpy(), (480, 854))
    np.testing.assert_array_equal(mask.cpu().numpy(), om)  # same arithmetic as the oracle: exact
    np.testing.assert_array_equal(small.cpu().numpy(), osm)


@pytest.mark.gpu
def test_label_resize_and_head_inputs_match_the_framework_ops():
    """ops.label_resize_nearest == F.interpolate(mask.float(), size, mode='nearest').int() (IntVOS.py:598-599) and
    ops.head_inputs == the eq / permute / cat assembly of the head's per-object channels (IntVOS.py:663-669), exactly."""
    import torch
    import torch.nn.functional as F
    from cvpr2020_manet_amd import ops
    g = torch.Generator(device="cuda").manual_seed(4)
    for (H, W, h, w, n_ids) in ((480, 854, 120, 214, 3), (33, 47, 9, 12, 5), (720, 1280, 180, 320, 6), (7, 5, 7, 5, 1),
                                (10, 10, 23, 31, 2)):
        mask = torch.randint(0, n_ids + 1, (1, 1, H, W), generator=g, device="cuda", dtype=torch.int64)
        want = F.interpolate(mask.float(), size=(h, w), mode="nearest").int()
        got = ops.label_resize_nearest(mask, (h, w))
        assert got.dtype == torch.int32 and torch.equal(got, want)
        gm = torch.rand(1, h, w, n_ids, 1, generator=g, device="cuda")
        lm = torch.rand(1, h, w, n_ids, 1, generator=g, device="cuda")
        lab = want[0].permute(1, 2, 0)  # [h, w, 1] as the module holds it
        ids = torch.arange(0, n_ids, dtype=torch.int32, device="cuda")
        prev = (lab.float() == ids.float()).unsqueeze(-1).permute(2, 3, 0, 1).float()
        ref = torch.cat((gm.squeeze(0).permute(2, 3, 0, 1), lm.squeeze(0).permute(2, 3, 0, 1), prev), 1)
        assert torch.equal(ops.head_inputs(gm, lm, lab, n_ids, (h, w)), ref)
    with pytest.raises(ValueError):
        ops.label_resize_nearest(torch.zeros(1, 1, 4, 4, device="cuda"), (2, 2))  # floating-point mask
    with pytest.raises(ValueError):
        ops.head_inputs(torch.zeros(5, device="cuda"), torch.zeros(5, device="cuda"), torch.zeros(2, 2, device="cuda"), 2, (2, 2))


@pytest.mark.gpu
def test_frame_begin_is_the_label_resize_plus_two_fills():
    """r5 manet_frame_begin: label_resize_nearest + the local map's pre-set + the distance weight (IntVOS.py:641) in ONE launch --
    the label equals the stand-alone op's, the fill and the scalar land where they should (and nowhere else), either may be absent;
    refusals for tensors it cannot write"""
    import torch
    from cvpr2020_manet_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    for (H, W, h, w, nfill) in ((480, 854, 120, 214, 120 * 214 * 3), (37, 53, 9, 14, 5), (64, 64, 16, 16, 16 * 16 * 9), (480, 854, 120, 214, 0)):
        mask = torch.randint(0, 5, (1, 1, H, W), generator=g, device=dev)
        want = ops.label_resize_nearest(mask, (h, w))
        buf = torch.full((nfill + 7,), -3.0, device=dev)
        tab = torch.zeros(6, 9, device=dev)
        got = ops.frame_begin(mask, (h, w), fill=buf[3:3 + nfill] if nfill else None, fill_value=1.0, scalar_dst=tab[4][2], scalar_value=1.0 / 3.0)
        assert torch.equal(got, want)
        assert bool((buf[3:3 + nfill] == 1.0).all()) and bool((buf[:3] == -3.0).all()) and bool((buf[3 + nfill:] == -3.0).all())
        assert float(tab[4][2]) == float(torch.tensor(1.0 / 3.0, dtype=torch.float32)) and int((tab != 0).sum()) == 1
        assert torch.equal(ops.frame_begin(mask, (h, w)), want)  # neither extra
    with pytest.raises(ValueError):
        ops.frame_begin(mask, (h, w), fill=torch.zeros(4, dtype=torch.float64, device=dev))
    with pytest.raises(ValueError):
        ops.frame_begin(mask, (h, w), scalar_dst=torch.zeros(2, device=dev))
    with pytest.raises(ValueError):
        ops.frame_begin(mask, (h, w), fill=torch.zeros(4, 4, device=dev)[:, 1])
    with pytest.raises(ValueError):
        ops.frame_begin(mask.float(), (h, w))
