"""The raw HIP ops return tensors without a grad_fn.  The reference's functions are differentiable and its
training drivers back-propagate through them (train_stage1.py:126-156), so an op without a backward must
refuse an input that requires grad instead of silently detaching it (VERDICT r1 weak #4 / ADVICE).
Runs on CPU: the guard fires before any device work."""
import pytest
import torch

from cvpr2020_manet_amd import ops


def _emb(requires_grad):
    return torch.rand(4, 5, 8, requires_grad=requires_grad)


@pytest.mark.parametrize("call", [
    lambda e: ops.local_dist(e, _emb(False), 2),
    lambda e: ops.local_dist(_emb(False), e, 2),
    lambda e: ops.normalize_merge_(e.reshape(-1)),
    lambda e: ops.PreparedBank(e, torch.zeros(4, 5, dtype=torch.int32), 2),
    lambda e: ops.dwconv7x7_bn_relu(e.permute(2, 0, 1)[None], torch.rand(8, 1, 7, 7)),
])
def test_ops_without_backward_refuse_grad_inputs(call):
    with pytest.raises(RuntimeError, match="requires grad"):
        call(_emb(True))


def test_guard_is_silent_without_grad_mode():
    # under no_grad the guard passes and the next check (no CPU fallback) is what fires
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="HIP device"):
            ops.local_dist(_emb(True), _emb(False), 2)
