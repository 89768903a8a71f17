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
