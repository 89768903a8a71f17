"""Drop-in counterpart of the reference's ``networks/IntVOS.py`` with the matching path on MI355X.

Same public names, signatures, return conventions and state-dict keys as the reference module, so
a ``test.py``-shaped driver keeps working after swapping the import:

  module functions   nearest_neighbor_features_per_object            (reference IntVOS.py:160-210)
                     local_pairwise_distances2                       (:266-315)
                     local_previous_frame_nearest_neighbor_features_per_object   (:345-434)
                     cross_correlate / local_pairwise_distances      (:212-265,:318-341, flag path)
  classes            IntVOS (:530-764), DynamicSegHead (:509-525), IntSegHead (:462-484),
                     _split_separable_conv2d (:488-506), _res_block (:440-458)

What differs, deliberately:
  * the distance / min / window arithmetic runs in hand-written HIP kernels behind the C ABI of
    ``include/manet_hip.h`` (``cvpr2020_manet_amd.ops``); nothing N x M or (2d+1)^2-unfolded is ever
    materialised, so ``n_chunks`` is accepted and ignored (the reference's results are
    chunk-invariant);
  * tensors stay on the device they arrive on (the reference calls ``.cuda()`` unconditionally,
    IntVOS.py:102,200); CPU tensors raise -- there is no CPU fallback;
  * normalisation + min-aggregation with the stored global map (:611-622) is fused into the
    matching kernel's epilogue;
  * segmentation heads and the encoder are stock PyTorch-ROCm modules (out of scope);
  * training: when grad mode is on and an embedding requires grad, the matching ops route through explicit
    torch.autograd.Functions (cvpr2020_manet_amd/autograd.py) so that ``loss.backward()`` of
    train_stage1.py:126-156 reaches the encoder exactly as it does through the reference's pure-PyTorch path.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..config import cfg as _default_cfg

# reference module constants (IntVOS.py:15-17)
USE_CORRELATION_COST = False
MODEL_UNFOLD = True
WRONG_LABEL_PADDING_DISTANCE = 1e20

# the flag namespace the module-level functions consult (the reference reads its global `cfg`);
# IntVOS(cfg, ...) re-binds it to the cfg it is given.
cfg = _default_cfg

# the heads' 1x1 convolutions with 256 output channels (inference fast path):
#   "f32"   : the exact fp32-MFMA kernel (ops.conv1x1_mfma; True is accepted as an alias) -- the default: the reference's
#             nn.Conv2d is an fp32 convolution;
#   "split" : split-bf16 MFMA kernel (ops.conv1x1_split: fp32 factors as hi + lo bf16 pieces, fp32 accumulation; error
#             <= 2^-16 relative per product -- between fp32 and the TF32 the reference's cuDNN path defaults to) -- the stage
#             becomes a stream over its activation (2.3x faster); opt-in;
#   "split3": the same kernel with THREE bf16 pieces per factor (hi + mid + lo = 24 significand bits) and six products per pair:
#             fp32-class results (~2^-23 relative per product, fp32 accumulation; not the fmaf chain's bits) at 6/16 of the
#             fp32 matrix pipe's time; opt-in;
#   False / "framework" : the framework's GEMM as in r2.
# This global is only the default of heads built OUTSIDE an IntVOS; a model carries its own mode
# (IntVOS(cfg, fe, pointwise=...) / cfg.MODEL_HEAD_POINTWISE), stamped on its _split_separable_conv2d blocks.
MFMA_POINTWISE = "f32"
_POINTWISE_MODES = {"f32": "f32", True: "f32", "split": "split", "split3": "split3", False: False, "framework": False}


def _pointwise_mode(block=None):
    mode = getattr(block, "_pw_mode", None) if block is not None else None
    if mode is None:
        mode = MFMA_POINTWISE
    return _POINTWISE_MODES[mode]

# arithmetic of the QK^T contraction used by the MODULE-LEVEL functions: "f32" (exact fp32 MFMA) | "bf16" | "bf16x3" |
# "bf16r".  An IntVOS instance carries its own (constructor argument / cfg.MODEL_MATCH_COMPUTE).
COMPUTE = "f32"


def set_cfg(new_cfg):
    global cfg
    cfg = new_cfg


class SynchronizedBatchNorm2d(nn.BatchNorm2d):
    """Name-compatible stand-in for the reference's vendored SyncBN (sync_batchnorm/batchnorm.py).
    In eval mode, and whenever the model is not wrapped in DataParallelWithCallback, the reference's
    SyncBN *is* F.batch_norm (batchnorm.py:48-53) -- no live caller wraps it."""


# --------------------------------------------------------------------------------------------------
# matching functions

def _n_ids_from(gt_ids, reference_labels_flat):
    """IntVOS.py:192-200: ids = arange(0, gt_ids+1); gt_ids None -> derived from the labels."""
    if gt_ids is None:
        return int(reference_labels_flat.max().item()) + 1  # torch.unique(...)[-1] == max
    if isinstance(gt_ids, torch.Tensor):
        return int(gt_ids.item()) + 1
    return int(gt_ids) + 1


def nearest_neighbor_features_per_object(reference_embeddings, query_embeddings, reference_labels,
                                         k_nearest_neighbors, gt_ids=None, n_chunks=100):
    """Distance to the nearest reference pixel per object (reference IntVOS.py:160-210).

    reference_embeddings [h_r, w_r, C], query_embeddings [h, w, C], reference_labels [h_r, w_r, 1]
    int.  Returns (nn_features float32 [1, h, w, n_ids, 1], ids int32 [n_ids]).
    `n_chunks` is ignored: the fused kernel tiles the bank through LDS instead of chunking queries.
    """
    assert reference_embeddings.size()[:2] == reference_labels.size()[:2]  # IntVOS.py:189
    h, w, _ = query_embeddings.size()
    labels_flat = reference_labels.reshape(-1)
    n_ids = _n_ids_from(gt_ids, labels_flat)
    if k_nearest_neighbors > 1 and cfg.TEST_MODE:
        # top-k counts bank rows: TEST_MODE drops unlabelled rows first (IntVOS.py:135-136)
        keep = labels_flat != -1
        reference_embeddings = reference_embeddings.reshape(-1, reference_embeddings.shape[-1])[keep]
        labels_flat = labels_flat[keep]
    out = ops.global_match(reference_embeddings, query_embeddings, labels_flat, n_ids,
                           k_nearest_neighbors=k_nearest_neighbors, compute=COMPUTE)
    ids = torch.arange(0, n_ids, dtype=torch.int32, device=out.device)
    return out.view(1, h, w, n_ids, 1), ids


def local_pairwise_distances2(x, y, max_distance=9):
    """Squared-L2 distances in a (2d+1)^2 window (reference IntVOS.py:266-315).
    x = query frame, y = previous frame, [h, w, C] -> [h, w, (2d+1)^2]."""
    return ops.local_dist(x, y, max_distance, downsample=bool(cfg.MODEL_LOCAL_DOWNSAMPLE))


def cross_correlate(x, y, max_distance=9):
    """Un-normalised windowed cross-correlation (reference IntVOS.py:318-341).

    The reference builds it from the third-party SpatialCorrelationSampler(kernel_size=1,
    patch_size=2d+1, stride=1, padding=0, dilation_patch=1), which is not vendored and has no pinned
    version: parity of this flag path is UNPINNED.  Its published semantics -- out[l=(dy,dx)]
    = sum_c x[c,y,x] * y[c,y+dy-d,x+dx-d], zero outside -- equal the correlation_package forward with
    pad=d, K=1, max_disp=d, s1=s2=1 times C, which is what runs here."""
    C = x.shape[-1]
    xs = x.permute(2, 0, 1).unsqueeze(0)
    ys = y.permute(2, 0, 1).unsqueeze(0)
    corr = ops.correlation_forward(xs, ys, max_distance, 1, max_distance, 1, 1) * float(C)
    return corr.squeeze(0).permute(1, 2, 0)


def local_pairwise_distances(x, y, max_distance=9):
    """Correlation formulation of the local distances (reference IntVOS.py:212-265); only used when
    USE_CORRELATION_COST is True.  See cross_correlate about parity."""
    def core(x, y):
        corr = cross_correlate(x, y, max_distance=max_distance)
        xs = torch.sum(x * x, 2, keepdim=True)
        ys = torch.sum(y * y, 2, keepdim=True)
        ones_ys = torch.ones_like(ys)
        ys = cross_correlate(ones_ys, ys, max_distance=max_distance)
        d = xs + ys - 2 * corr
        boundary = torch.eq(cross_correlate(ones_ys, ones_ys, max_distance=max_distance), 0)
        return torch.where(boundary, torch.full_like(d, float("inf")), d)

    if cfg.MODEL_LOCAL_DOWNSAMPLE:
        ori_h, ori_w, _ = x.size()
        x = F.avg_pool2d(x.permute(2, 0, 1).unsqueeze(0), (2, 2), (2, 2)).squeeze(0).permute(1, 2, 0)
        y = F.avg_pool2d(y.permute(2, 0, 1).unsqueeze(0), (2, 2), (2, 2)).squeeze(0).permute(1, 2, 0)
        d = core(x.contiguous(), y.contiguous())
        d = (torch.sigmoid(d) - 0.5) * 2
        d = F.interpolate(d.permute(2, 0, 1).unsqueeze(0), size=(ori_h, ori_w), mode="bilinear",
                          align_corners=True)
        return d.squeeze(0).permute(1, 2, 0)
    return core(x, y)


def local_previous_frame_nearest_neighbor_features_per_object(prev_frame_embedding, query_embedding,
                                                              prev_frame_labels, gt_ids,
                                                              max_distance=12):
    """Nearest-neighbour distance per object inside a local window of the previous frame
    (reference IntVOS.py:345-434, unfold path).  gt_ids: int tensor [n_ids] = 0..n_ids-1.
    Returns float32 [1, h, w, n_ids, 1]."""
    h, w = prev_frame_embedding.size()[:2]
    n_ids = int(gt_ids.size(0))
    out = ops.local_match(prev_frame_embedding, query_embedding, prev_frame_labels, n_ids,
                          max_distance=max_distance, downsample=bool(cfg.MODEL_LOCAL_DOWNSAMPLE))
    return out.view(1, h, w, n_ids, 1)


# --------------------------------------------------------------------------------------------------
# heads (stock PyTorch-ROCm modules; same parameter names as the reference)

class _res_block(nn.Module):  # reference IntVOS.py:440-458
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.conv1 = nn.Conv2d(in_dim, out_dim, kernel_size=3, stride=1, padding=1)
        self.relu1 = nn.ReLU()
        self.bn1 = SynchronizedBatchNorm2d(out_dim, momentum=cfg.TRAIN_BN_MOM)
        self.conv2 = nn.Conv2d(out_dim, out_dim, kernel_size=3, stride=1, padding=1)
        self.relu2 = nn.ReLU()
        self.bn2 = SynchronizedBatchNorm2d(out_dim, momentum=cfg.TRAIN_BN_MOM)

    def forward(self, x):
        res = x
        x = self.relu1(self.bn1(self.conv1(x)))
        x = self.relu2(self.bn2(self.conv2(x)))
        x += res
        return x


class IntSegHead(nn.Module):  # reference IntVOS.py:462-484
    def __init__(self, in_dim=None, emb_dim=None):
        super().__init__()
        in_dim = cfg.MODEL_SEMANTIC_EMBEDDING_DIM + 3 if in_dim is None else in_dim
        emb_dim = cfg.MODEL_HEAD_EMBEDDING_DIM if emb_dim is None else emb_dim
        self.conv1 = nn.Conv2d(in_dim, emb_dim, kernel_size=7, stride=1, padding=3)
        self.bn1 = SynchronizedBatchNorm2d(emb_dim, momentum=cfg.TRAIN_BN_MOM)
        self.relu1 = nn.ReLU(True)
        self.res1 = _res_block(emb_dim, emb_dim)
        self.res2 = _res_block(emb_dim, emb_dim)
        self.conv2 = nn.Conv2d(256, emb_dim, kernel_size=3, stride=1, padding=1)
        self.bn2 = SynchronizedBatchNorm2d(emb_dim, momentum=cfg.TRAIN_BN_MOM)
        self.relu2 = nn.ReLU(True)
        self.conv3 = nn.Conv2d(emb_dim, 1, 1, 1)

    def forward(self, x):
        x = self.relu1(self.bn1(self.conv1(x)))
        x = self.res2(self.res1(x))
        x = self.relu2(self.bn2(self.conv2(x)))
        return self.conv3(x)


class _split_separable_conv2d(nn.Module):  # reference IntVOS.py:488-506
    def __init__(self, in_dim, out_dim, kernel_size=7):
        super().__init__()
        self.conv1 = nn.Conv2d(in_dim, in_dim, kernel_size=kernel_size, stride=1,
                               padding=int((kernel_size - 1) / 2), groups=in_dim)
        self.relu1 = nn.ReLU(True)
        self.bn1 = SynchronizedBatchNorm2d(in_dim, momentum=cfg.TRAIN_BN_MOM)
        self.conv2 = nn.Conv2d(in_dim, out_dim, kernel_size=1, stride=1)
        self.relu2 = nn.ReLU(True)
        self.bn2 = SynchronizedBatchNorm2d(out_dim, momentum=cfg.TRAIN_BN_MOM)
        nn.init.kaiming_normal_(self.conv1.weight, mode="fan_out", nonlinearity="relu")
        nn.init.kaiming_normal_(self.conv2.weight, mode="fan_out", nonlinearity="relu")

    def _fast(self, x):
        """fused HIP depthwise-7x7 + BN + ReLU (ops.dwconv7x7_bn_relu) is usable: inference on the GPU"""
        return (not self.training and x.is_cuda and not torch.is_grad_enabled()
                and self.conv1.kernel_size == (7, 7) and x.dtype == torch.float32)

    def _folded(self, cs=0):
        """Inference constants of the block, computed once and reused until a parameter or running statistic changes
        (r2 trace of examples/propagate_clip.py: re-folding both BatchNorms per call was 52 of the 76 kernel launches
        of a propagated frame).  Returns (scale1, shift1) of eval-mode bn1 and conv2 with eval-mode bn2 folded into
        its weights -- bn2(conv2(x)) == conv2'(x) -- the latter also split at input channel `cs` (forward_shared)."""
        src = [self.bn1.weight, self.bn1.bias, self.bn1.running_mean, self.bn1.running_var, self.bn2.weight,
               self.bn2.bias, self.bn2.running_mean, self.bn2.running_var, self.conv2.weight, self.conv2.bias]
        key = (cs, _pointwise_mode(self)) + tuple((t.data_ptr(), t._version) if t is not None else None for t in src)
        hit = getattr(self, "_fold_cache", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        with torch.no_grad():
            scale1, shift1 = ops.fold_bn(self.bn1)
            scale2, shift2 = ops.fold_bn(self.bn2)
            w2 = (self.conv2.weight.detach().float() * scale2[:, None, None, None]).contiguous()
            b2 = (self.conv2.bias.detach().float() * scale2 + shift2 if self.conv2.bias is not None else shift2).contiguous()
            val = {"scale1": scale1.contiguous(), "shift1": shift1.contiguous(), "w2": w2, "b2": b2,
                   "b2_zero": torch.zeros_like(b2),  # (made here: exists before a driver forks its streams, prepare_bank)
                   "w2_shared": w2[:, :cs].contiguous(), "w2_object": w2[:, cs:].contiguous()}
            mode = _pointwise_mode(self)
            if self.conv2.out_channels == ops.PW_COUT and mode and w2.is_cuda:  # the weight transposed [Cin, Cout]
                w2t = w2.reshape(w2.shape[0], w2.shape[1]).t().contiguous()
                if mode in ("split", "split3"):  # ops.conv1x1_split: packed hi / (mid /) lo operand images
                    np_ = 3 if mode == "split3" else 2
                    val["sw"] = ops.SplitWeight(w2t, pieces=np_)
                    if 0 < cs < w2t.shape[0]:
                        val["sw_shared"], val["sw_object"] = ops.SplitWeight(w2t[:cs], pieces=np_), ops.SplitWeight(w2t[cs:], pieces=np_)
                else:  # ops.conv1x1_mfma
                    val["w2t"], val["w2t_shared"], val["w2t_object"] = w2t, w2t[:cs].contiguous(), w2t[cs:].contiguous()
            val["mode"] = mode
        object.__setattr__(self, "_fold_cache", (key, val))  # plain attribute: not a buffer, not in the state dict
        return val

    def _pointwise(self, y, k, which, bias, relu):
        """bn2(conv2(y)) [+ relu2] on the depthwise stage's output: the fp32-MFMA 1x1 kernel when the shapes allow (256 output
        channels, h*w a multiple of 4), else the framework's convolution with the same folded weights"""
        skey = {"all": "sw", "shared": "sw_shared", "object": "sw_object"}[which]
        if skey in k and ops.conv1x1_split_ok(y, self.conv2.out_channels):
            b2 = k["b2"] if bias else k["b2_zero"]
            return ops.conv1x1_split(y, k[skey], b2, relu_out=relu)
        wkey = {"all": "w2t", "shared": "w2t_shared", "object": "w2t_object"}[which]
        if wkey in k and ops.conv1x1_mfma_ok(y, self.conv2.out_channels):
            b2 = k["b2"] if bias else k["b2_zero"]
            return ops.conv1x1_mfma(y, k[wkey], b2, relu_out=relu)
        w = {"all": "w2", "shared": "w2_shared", "object": "w2_object"}[which]
        z = F.conv2d(y, k[w], k["b2"] if bias else None)
        return z.relu_() if relu else z

    def forward(self, x, relu_in=False, defer_relu=False, head=None):
        """relu_in / defer_relu (inference fast path only, used by DynamicSegHead): the block's last ReLU is left to the
        NEXT block, whose fused depthwise kernel reads its input through max(x, 0) -- one elementwise pass over the
        activation less per block, same values.  head = (weight, bias) of DynamicSegHead's output layer: fused into this
        block's 1x1 kernel when it runs on the MFMA path (returns [B, 1, h, w]), else applied behind it."""
        if self._fast(x):
            k = self._folded()
            x = ops.dwconv7x7_bn_relu(x, self.conv1.weight, self.conv1.bias, scale=k["scale1"], shift=k["shift1"],
                                      relu_in=relu_in)
            if head is not None:
                if "sw" in k and ops.conv1x1_split_ok(x, self.conv2.out_channels):
                    return ops.conv1x1_split(x, k["sw"], k["b2"], head_weight=head[0], head_bias=head[1])
                if "w2t" in k and ops.conv1x1_mfma_ok(x, self.conv2.out_channels):
                    return ops.conv1x1_mfma(x, k["w2t"], k["b2"], head_weight=head[0], head_bias=head[1])
                return ops.relu_conv1x1_c1(self._pointwise(x, k, "all", True, False), head[0], head[1])
            return self._pointwise(x, k, "all", True, not defer_relu)
        assert not relu_in and not defer_relu
        x = self.relu1(self.bn1(self.conv1(x)))
        x = self.relu2(self.bn2(self.conv2(x)))
        return x

    def forward_shared(self, shared, per_object, defer_relu=False, memo=None):
        """The block applied to cat([shared repeated n times, per_object], 1) without building that tensor
        (IntVOS.py:665-670 repeats the C-channel embedding once per object): the depthwise stage and the
        1x1 stage are linear in the channel groups, so the shared group is processed once.
        shared [1, Cs, h, w], per_object [n, Cp, h, w], Cs + Cp == in_dim.
        memo (r5): a one-slot holder {"k": folded constants, "term": tensor} of the SHARED half's contribution
        conv2'(dw(shared)) -- it depends on the frame's embedding and the layer's parameters only, not on objects, labels or
        the interaction round, so a caller that sees the same frame again (every round of a session walks the same clip)
        hands the holder back and the shared depthwise + 1x1 launches are skipped.  Valid while `k` is the SAME folded-constant
        object (a parameter change re-folds and drops it)."""
        cs = shared.shape[1]
        k = self._folded(cs)
        scale, shift = k["scale1"], k["shift1"]
        w1, b1 = self.conv1.weight, self.conv1.bias
        p1 = ops.dwconv7x7_bn_relu(per_object, w1[cs:], b1[cs:], scale=scale[cs:], shift=shift[cs:])
        split = "sw_object" in k and ops.conv1x1_split_ok(p1, self.conv2.out_channels)
        mfma = not split and "w2t_object" in k and ops.conv1x1_mfma_ok(p1, self.conv2.out_channels)
        term = None
        stamp = _memo_stamp(self, k)
        if memo is not None and memo.get("k") is k and memo.get("stamp") == stamp and (split or mfma):
            term = memo.get("term")
        if term is None:
            s1 = ops.dwconv7x7_bn_relu(shared, w1[:cs], b1[:cs], scale=scale[:cs], shift=shift[:cs])
            if split:
                term = ops.conv1x1_split(s1, k["sw_shared"], k["b2_zero"])
            elif mfma:
                term = ops.conv1x1_mfma(s1, k["w2t_shared"], k["b2_zero"])
            if memo is not None and term is not None:
                memo["k"], memo["term"], memo["stamp"] = k, term, stamp
        if split:
            # the shared half once, then the per-object half with it added in the epilogue (no broadcast-add pass)
            return ops.conv1x1_split(p1, k["sw_object"], k["b2"], relu_out=not defer_relu, add=term)
        if mfma:  # the same form on the fp32 matrix pipe
            return ops.conv1x1_mfma(p1, k["w2t_object"], k["b2"], relu_out=not defer_relu, add=term)
        y = self._pointwise(p1, k, "object", True, False)
        y += self._pointwise(s1, k, "shared", False, False)  # broadcast over the objects
        return y if defer_relu else y.relu_()


def _memo_stamp(layer, k):
    """what the memoised shared-half term conv2'(dw(shared)) depends on besides the frame: the folded constants `k` (bn1, bn2,
    conv2 -- compared by identity: a change re-folds) AND the depthwise layer's own parameters, which `_folded`'s key does not
    cover (an in-place change of conv1 alone left a stale term in use)"""
    w1, b1 = layer.conv1.weight, layer.conv1.bias
    try:
        return (w1.data_ptr(), w1._version, None if b1 is None else (b1.data_ptr(), b1._version))
    except RuntimeError:  # inference tensors: no version counter -- never equal to a stored stamp
        return object()


def _layer1_fused(layer, shared, global_map, local_map, labels, n_ids, size, memo=None):
    """`layer`.forward_shared on head_inputs(global_map, local_map, labels) in TWO launches less (r5): the shared-embedding half
    as in forward_shared (memoised per frame), then ops.head_layer1_object -- input assembly + the per-object channels' depthwise
    stage + their 1x1 + the shared term + ReLU in one launch.  Returns None when the layer is not on the exact-fp32 MFMA path (the
    caller then takes the general route)."""
    cs = shared.shape[1]
    k = layer._folded(cs)
    if k.get("mode") != "f32" or "w2t_object" not in k or layer.conv1.in_channels != cs + 3 or layer.conv2.out_channels != ops.PW_COUT:
        return None
    # (eligibility from the shapes, BEFORE any launch: the depthwise stage keeps [1, cs, h, w] -- ADVICE r5)
    if shared.dim() != 4 or shared.dtype != torch.float32 or not ops.conv1x1_mfma_ok(shared, layer.conv2.out_channels):
        return None
    w1, b1 = layer.conv1.weight, layer.conv1.bias
    stamp = _memo_stamp(layer, k)
    term = memo.get("term") if (memo is not None and memo.get("k") is k and memo.get("stamp") == stamp) else None
    if term is None:
        s1 = ops.dwconv7x7_bn_relu(shared, w1[:cs], b1[:cs], scale=k["scale1"][:cs], shift=k["shift1"][:cs])
        term = ops.conv1x1_mfma(s1, k["w2t_shared"], k["b2_zero"])
        if memo is not None:
            memo["k"], memo["term"], memo["stamp"] = k, term, stamp
    return ops.head_layer1_object(global_map, local_map, labels, n_ids, size, w1[cs:], None if b1 is None else b1[cs:],
                                  k["scale1"][cs:], k["shift1"][cs:], k["w2t_object"], k["b2"], term, relu_out=True)


class DynamicSegHead(nn.Module):  # reference IntVOS.py:509-525
    def __init__(self, in_dim=None, embed_dim=None, kernel_size=1):
        super().__init__()
        in_dim = cfg.MODEL_SEMANTIC_EMBEDDING_DIM + 3 if in_dim is None else in_dim
        embed_dim = cfg.MODEL_HEAD_EMBEDDING_DIM if embed_dim is None else embed_dim
        self.layer1 = _split_separable_conv2d(in_dim, embed_dim)
        self.layer2 = _split_separable_conv2d(embed_dim, embed_dim)
        self.layer3 = _split_separable_conv2d(embed_dim, embed_dim)
        self.layer4 = _split_separable_conv2d(embed_dim, embed_dim)
        self.conv = nn.Conv2d(embed_dim, 1, 1, 1)
        nn.init.kaiming_normal_(self.conv.weight, mode="fan_out", nonlinearity="relu")

    def _tail(self, x):
        """layers 2-4 + the 1x1 output conv on the inference fast path; x = layer1's output.  r5: every block's last ReLU
        is applied in the epilogue of its own 1x1 kernel, where it is free -- r2-r4 left it to the NEXT block's depthwise
        kernel (`relu_in`, from the days the 1x1 stage was the framework's GEMM and a ReLU meant a pass): 60 v_max per
        staged tile less (0.5 us of 47 at [3,256,120,214]: that kernel hides them).  Same values either way.  (`relu_in` /
        `defer_relu` remain for blocks whose 1x1 stage falls back to the framework's convolution.)"""
        x = self.layer2(x)
        x = self.layer3(x)
        if self.conv.kernel_size == (1, 1) and self.conv.out_channels == 1:
            # output layer fused with layer4's ReLU -- into the epilogue of layer4's own 1x1 kernel when that runs on the
            # MFMA path (layer4's activation is never written), else one pass over it (ops.relu_conv1x1_c1)
            return self.layer4(x, defer_relu=True, head=(self.conv.weight, self.conv.bias))
        return self.conv(self.layer4(x))

    def forward(self, x):
        if self.layer1._fast(x):
            return self._tail(self.layer1(x))
        return self.conv(self.layer4(self.layer3(self.layer2(self.layer1(x)))))

    def forward_shared(self, shared, per_object, memo=None):
        """forward(cat([shared.repeat(n,1,1,1), per_object], 1)) without materialising the input; memo: see
        _split_separable_conv2d.forward_shared"""
        return self._tail(self.layer1.forward_shared(shared, per_object, memo=memo))


def _run_head(head, embedding_chw, per_object, memo=None):
    """head(cat([embedding repeated per object, per_object], 1)) (IntVOS.py:665-671, :741-758).  In
    inference on the GPU a DynamicSegHead takes the shared-embedding route (no repeat / cat of the C-channel
    embedding, depthwise stage of those channels computed once); otherwise the reference's literal form.
    memo: the frame's holder of layer 1's shared-half term (IntVOS._head_memo), or None."""
    n = per_object.shape[0]
    if embedding_chw.dtype != torch.float32:  # 2-byte embeddings (MODEL_EMB_DTYPE bf16): the head computes in fp32
        embedding_chw = embedding_chw.float()
    if (isinstance(head, DynamicSegHead) and not head.training and not torch.is_grad_enabled()
            and embedding_chw.is_cuda and embedding_chw.dtype == torch.float32):
        return head.forward_shared(embedding_chw.unsqueeze(0), per_object, memo=memo)
    return head(torch.cat((embedding_chw.unsqueeze(0).repeat((n, 1, 1, 1)), per_object), 1))


# --------------------------------------------------------------------------------------------------
_ids_cache = {}


def _obj_ids(n_ids, device):
    """torch.arange(0, n_ids, int32) on `device` (IntVOS.py:573 builds it per frame: one launch), made once"""
    key = (int(n_ids), str(device))
    t = _ids_cache.get(key)
    if t is None:
        t = _ids_cache[key] = torch.arange(0, n_ids, dtype=torch.int32, device=device)
    return t


MAX_CLIP_FRAMES = 104       # hard-coded clip length of the reference's memories (IntVOS.py:617,645)
MAX_INTERACTIONS = 9        # IntVOS.py:641,645
MAX_CACHED_FRAMES = 2 * MAX_CLIP_FRAMES + 8  # most prepared per-frame operands a model ever keeps: 17 MB each at 480p, plus --
                                             # once a head has seen the frame -- its 26 MB layer-1 term (capped: head_memo_bytes_cap)
DEFAULT_CACHED_FRAMES = 8   # ... and what it keeps unless the driver prepared a clip up front (prepare_clip /
                            # extract_feature(packed=True)): test.py:259's cur -> prev hand-over, the annotated frames of a few
                            # rounds (136 MB at 480p; r3 kept up to 216 frames of every tensor that ever came by: ADVICE r3)
MAX_CACHED_BANKS = 2        # prepared memory banks kept per model (one per sequence name)
DEFAULT_HEAD_MEMO_MB = 8192  # the heads' memoised layer-1 shared-half terms kept on cached frames: 26 MB per 480p frame and head
DEFAULT_LOCAL_VOLUME_CACHE_MB = 8192  # stored local-match volumes (prepare_local_volumes): 25.8 MB per 480p frame pair at d = 12,
                                      # two directions x (F - 1) pairs per clip (5.1 GB for 100 frames); LRU beyond the cap
_EMB_DTYPES = {"f32": torch.float32, "fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
               "bfloat16": torch.bfloat16, torch.float32: torch.float32, torch.bfloat16: torch.bfloat16}


class IntVOS(nn.Module):
    """reference IntVOS.py:530-764: same constructor, methods, dict conventions, state-dict keys."""

    def __init__(self, cfg, feature_extracter, compute=None, emb_dtype=None, pointwise=None, cache_frames=None):
        """cfg, feature_extracter: as the reference.  The rest is optional and this implementation's only (default: the
        cfg's MODEL_MATCH_COMPUTE / MODEL_EMB_DTYPE / MODEL_HEAD_POINTWISE / MODEL_CACHE_FRAMES when it has them, else
        "f32" / "f32" / "f32" / True):
          compute       arithmetic of the global match: "f32" exact | "bf16" | "bf16x3" | "bf16r"
          emb_dtype     storage type of extract_feature's output: "f32" | "bf16" (the matching kernels then read 2-byte
                        embeddings end to end; the heads widen them)
          pointwise     the heads' 256-channel 1x1 layers in inference: "f32" exact fp32-MFMA kernel | "split" split-bf16
                        MFMA kernel (<= 2^-16 relative per product, 2.3x faster) | "split3" three-piece split (fp32-class:
                        ~2^-23 per product) | "framework" the framework's GEMM
          cache_frames  keep prepared per-frame operands keyed on the embedding tensor's identity (_prepared_frame).
                        False: every call prepares afresh -- REQUIRED when embeddings are rewritten in place without
                        torch noticing (HIP-graph replay of the encoder into a static buffer, `.data` writes)"""
        super().__init__()
        set_cfg(cfg)
        self.cfg = cfg
        pw = pointwise if pointwise is not None else getattr(cfg, "MODEL_HEAD_POINTWISE", "f32")
        if pw not in _POINTWISE_MODES:
            raise ValueError("pointwise=%r ('f32', 'split', 'split3' or 'framework')" % (pw,))
        self.pointwise = pw
        cf = cache_frames if cache_frames is not None else getattr(cfg, "MODEL_CACHE_FRAMES", True)
        self.cache_frames = bool(cf)
        self._frame_cache_cap = DEFAULT_CACHED_FRAMES
        self.compute = compute if compute is not None else getattr(cfg, "MODEL_MATCH_COMPUTE", "f32")
        if self.compute not in ops.COMPUTE:
            raise ValueError("compute=%r (one of %s)" % (self.compute, sorted(ops.COMPUTE)))
        ed = emb_dtype if emb_dtype is not None else getattr(cfg, "MODEL_EMB_DTYPE", "f32")
        if ed not in _EMB_DTYPES:
            raise ValueError("emb_dtype=%r ('f32' or 'bf16')" % (ed,))
        self.emb_dtype = _EMB_DTYPES[ed]
        self.feature_extracter = feature_extracter  # embedding extractor (out of scope; any module)
        self.feature_extracter.cls_conv = nn.Sequential()
        self.feature_extracter.upsample4 = nn.Sequential()
        self.semantic_embedding = None
        self.seperate_conv = nn.Conv2d(cfg.MODEL_ASPP_OUTDIM, cfg.MODEL_ASPP_OUTDIM, kernel_size=3, stride=1,
                                       padding=1, groups=cfg.MODEL_ASPP_OUTDIM)
        self.bn1 = SynchronizedBatchNorm2d(cfg.MODEL_ASPP_OUTDIM, momentum=cfg.TRAIN_BN_MOM)
        self.relu1 = nn.ReLU(True)
        self.embedding_conv = nn.Conv2d(cfg.MODEL_ASPP_OUTDIM, cfg.MODEL_SEMANTIC_EMBEDDING_DIM, 1, 1)
        self.relu2 = nn.ReLU(True)
        self.bn2 = SynchronizedBatchNorm2d(cfg.MODEL_SEMANTIC_EMBEDDING_DIM, momentum=cfg.TRAIN_BN_MOM)
        # the same modules registered twice -> aliased state-dict keys, as in the reference (:543)
        self.semantic_embedding = nn.Sequential(*[self.seperate_conv, self.bn1, self.relu1, self.embedding_conv,
                                                  self.bn2, self.relu2])
        for m in self.semantic_embedding:
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._dist_mirror = {}             # seq_name -> [its local_map_dist table, its version after this module's last write,
                                           #              {(frame, round): weight}]: _mirror_of
        self._bank_cache = OrderedDict()   # seq_name -> (identity key, ops.PreparedBank, keyed tensors): _prepared_bank
        self._frame_cache = OrderedDict()  # identity key of a [C,h,w] embedding -> ops.PreparedFrame: _prepared_frame
        self._memo_bytes = 0               # bytes of the heads' memoised shared-half terms on the frame entries (_head_memo)
        self.head_memo_bytes_cap = int(getattr(cfg, "MODEL_HEAD_MEMO_MB", DEFAULT_HEAD_MEMO_MB)) << 20
        self._vol_cache = OrderedDict()    # (key of the previous frame, key of the current frame) -> [volume, keep-alive tensors]
        self._vol_cache_bytes = 0
        self.local_volume_cache_bytes = int(getattr(cfg, "MODEL_LOCAL_VOLUME_CACHE_MB", DEFAULT_LOCAL_VOLUME_CACHE_MB)) << 20
        self.local_volume_lazy = bool(getattr(cfg, "MODEL_LOCAL_VOLUME_LAZY", False))
        self.dynamic_seghead = DynamicSegHead()  # propagation head
        if cfg.MODEL_USEIntSeg:
            self.inter_seghead = IntSegHead(in_dim=cfg.MODEL_SEMANTIC_EMBEDDING_DIM + 3)
        else:
            self.inter_seghead = DynamicSegHead(in_dim=cfg.MODEL_SEMANTIC_EMBEDDING_DIM + 2)  # interaction head
        for m in self.modules():  # this model's 1x1 mode (not a process-wide switch: two models may differ)
            if isinstance(m, _split_separable_conv2d):
                object.__setattr__(m, "_pw_mode", self.pointwise)

    def _prepared_bank(self, seq_name, ref_emb_chw, ref_label, ref_emb_hwc, ref_lab_flat, n_ids):
        """The sorted / packed memory bank of the annotated frame, reused while the caller keeps passing the SAME
        embedding and scribble tensors (identity = storage pointer, shape, strides and torch's in-place version
        counter): test.py:237-259 / :276-295 propagate a whole clip against one annotated frame, so the bank is
        sorted and packed once per interaction instead of once per frame.  One bank per sequence name, the
        MAX_CACHED_BANKS most recently used sequences.  (Writes that bypass torch's version counter -- `.data`, raw
        pointers -- are invisible to the key: call invalidate_caches() after such a write.)"""
        hit = self._bank_cache.get(seq_name)
        try:
            key = (ref_emb_chw.data_ptr(), tuple(ref_emb_chw.shape), tuple(ref_emb_chw.stride()), ref_emb_chw._version,
                   ref_emb_chw.dtype, ref_label.data_ptr(), tuple(ref_label.shape), ref_label._version, ref_label.dtype,
                   n_ids, self.compute, bool(self.cfg.TEST_MODE))
        except RuntimeError:
            # inference tensors carry no version counter: nothing trustworthy to key on -- sort / pack afresh (into the
            # previous bank's workspace) on every call
            bank = ops.PreparedBank(ref_emb_hwc, ref_lab_flat() if callable(ref_lab_flat) else ref_lab_flat, n_ids,
                                    compute=self.compute, reuse=hit[1] if hit is not None else None)
            self._bank_cache[seq_name] = (None, bank, ref_emb_chw, ref_label)
            return bank
        if hit is not None and hit[0] == key:
            self._bank_cache.move_to_end(seq_name)
            return hit[1]
        # (ref_lab_flat may be a callable: the caller converts the labels only when the bank really is rebuilt)
        bank = ops.PreparedBank(ref_emb_hwc, ref_lab_flat() if callable(ref_lab_flat) else ref_lab_flat, n_ids,
                                compute=self.compute, reuse=hit[1] if hit is not None else None)  # (the stale bank's workspace is taken over)
        # keep the keyed tensors alive so that their storage pointers cannot be recycled under the key
        self._bank_cache[seq_name] = (key, bank, ref_emb_chw, ref_label)
        self._bank_cache.move_to_end(seq_name)
        while len(self._bank_cache) > MAX_CACHED_BANKS:
            self._bank_cache.popitem(last=False)
        return bank

    def _mirror_of(self, seq_name, dist_tab):
        """Host mirror of the weights THIS module wrote into `dist_tab` (the caller's local_map_dist_dic[seq_name]), or None
        when the table cannot be tracked.  Valid only for the very tensor object it was made for (held here, so its address
        cannot be recycled: test.py:121-124,313-314 drops the dicts after a session and builds fresh ones for the next session
        on the same sequence name -- ADVICE r4: a data_ptr key took the new table for the old one) and only while the table's
        version counter still is what this module's last write left (an external write -- zero_(), a caller's own assignment
        -- starts the mirror afresh: the comparison then asks the device, as the reference does)."""
        try:
            ver = dist_tab._version
        except RuntimeError:  # inference tensors carry no version counter
            self._dist_mirror.pop(seq_name, None)
            return None
        m = self._dist_mirror.get(seq_name)
        if m is None or m[0] is not dist_tab or m[1] != ver:
            m = self._dist_mirror[seq_name] = [dist_tab, ver, {}]
        return m

    # ---- per-frame operands (the producer side of the path) ------------------------------------
    def _frame_key(self, emb_chw, d):
        """identity of an embedding tensor, or None when it has none that can be trusted: inference tensors carry no
        version counter (torch raises on `._version`) -- such frames are prepared afresh on every call"""
        try:
            ver = emb_chw._version
        except RuntimeError:
            return None
        return (emb_chw.data_ptr(), tuple(emb_chw.shape), tuple(emb_chw.stride()), ver, emb_chw.dtype, self.compute, d)

    def _local_radius(self):
        """window radius the pooled planes are padded for; -1 (no plane) when the fused local kernel does not apply"""
        d = int(self.cfg.MODEL_MAX_LOCAL_DISTANCE)
        return d if (self.cfg.MODEL_LOCAL_DOWNSAMPLE and 0 <= d <= 12) else -1

    def _prepared_frame(self, emb_chw, preset=None):
        """ops.PreparedFrame of one [C,h,w] embedding: the query operand image of the global match + the pooled plane of
        the local match, made by ONE launch from one read of the embedding, and reused for as long as the caller keeps
        passing the same tensor -- test.py:259 `prev_embedding = current_embedding` makes every frame the previous
        frame of the next step, and every interaction round walks the same `embedding_memory`.  Returns
        (frame, preset_done).  Inside a HIP-graph capture nothing is cached (a replay sees new contents in the same
        buffers), the prepare launch is part of the graph."""
        d = self._local_radius()
        key = None
        if self.cache_frames and not torch.cuda.is_current_stream_capturing():
            key = self._frame_key(emb_chw, d)
        hit = self._frame_cache.get(key) if key is not None else None
        if hit is not None:
            self._frame_cache.move_to_end(key)
            return hit, False
        frame = ops.prepare_frames(emb_chw, compute=self.compute, max_distance=d, preset=preset)
        if key is not None:
            frame.keep = emb_chw  # the key holds a storage pointer: keep the tensor alive with the entry
            self._store_frame(key, frame)
            self._trim_frames()
        return frame, preset is not None

    def _store_frame(self, key, frame):
        old = self._frame_cache.get(key)
        if old is not None:
            self._memo_bytes -= old.__dict__.get("memo_bytes", 0)
        self._frame_cache[key] = frame

    def _trim_frames(self):
        while len(self._frame_cache) > self._frame_cache_cap:
            _, old = self._frame_cache.popitem(last=False)
            self._memo_bytes -= old.__dict__.get("memo_bytes", 0)

    def _head_memo(self, emb_chw, head):
        """Holder of `head`.layer1's shared-half term for this frame (r5), or None.  It lives on the frame's entry of the
        frame cache -- same identity key (storage pointer, shape, strides, version counter), same lifetime, same opt-in: with
        the default 8-frame cache a clip's frames are evicted before the next round comes back to them; a driver that took
        `prepare_clip` / `extract_feature(packed=True)` keeps every frame and from the second round on skips two launches per
        propagated frame (the shared depthwise + its 1x1: 35 us of a 1 260 us frame at 480p; 26 MB per 480p frame kept).
        Nothing is kept inside a HIP-graph capture or for tensors without a version counter."""
        if not self.cache_frames or torch.cuda.is_current_stream_capturing():
            return None
        key = self._frame_key(emb_chw, self._local_radius())
        fr = self._frame_cache.get(key) if key is not None else None
        if fr is None:
            return None
        memos = fr.__dict__.setdefault("head_memos", {})
        holder = memos.get(id(head))
        if holder is None:
            # a term is [256, h, w] fp32 (26 MB at 480p) per frame and head: bounded by `head_memo_bytes_cap` --
            # beyond it the frame simply recomputes its shared half (two launches) every round
            per = 4 * ops.PW_COUT * int(emb_chw.shape[-2]) * int(emb_chw.shape[-1])
            if self._memo_bytes + per > self.head_memo_bytes_cap:
                return None
            holder = memos[id(head)] = {}
            fr.__dict__["memo_bytes"] = fr.__dict__.get("memo_bytes", 0) + per
            self._memo_bytes += per
        return holder

    def drop_head_memos(self):
        """forget the heads' memoised layer-1 terms (the frames' operands stay)"""
        for fr in self._frame_cache.values():
            fr.__dict__.pop("head_memos", None)
            fr.__dict__.pop("memo_bytes", None)
        self._memo_bytes = 0

    def prepare_head_terms(self, embeddings, head=None, batch=8):
        """Optional (r6), for drivers that prepared a clip (`prepare_clip` / `extract_feature(packed=True)`): the label-independent
        part of the propagation head -- layer 1's shared-embedding half, conv2'(relu(bn1(dw7x7(embedding)))) (IntVOS.py:488-507 on
        the 100 embedding channels of the head input, :665-670) -- for every frame of `embeddings` [F, C, h, w], `batch` frames per
        launch, stored where `prop_seghead` looks for it (the frames' memo holders: `_head_memo`).  Without this call each frame
        computes its term the first time a head sees it (two launches, 36 us at 480p, on the sequential chain of a sequence's first
        round).  The same kernels on the same planes: the same bits.  Returns the number of frames whose term is stored."""
        head = self.dynamic_seghead if head is None else head
        if (not isinstance(head, DynamicSegHead) or head.training or self.training or not self.cache_frames
                or torch.cuda.is_current_stream_capturing() or torch.is_grad_enabled()):
            return 0
        F_ = int(embeddings.shape[0])
        if F_ == 0:
            return 0
        layer, cs = head.layer1, int(embeddings.shape[1])
        k = layer._folded(cs)
        if (k.get("mode") != "f32" or "w2t_shared" not in k or layer.conv1.in_channels != cs + 3
                or layer.conv2.out_channels != ops.PW_COUT or layer.conv1.kernel_size != (7, 7)):
            return 0
        stamp = _memo_stamp(layer, k)
        w1, b1 = layer.conv1.weight, layer.conv1.bias
        todo, done = [], 0
        for i in range(F_):
            e = embeddings[i]
            if not e.is_cuda or (e.shape[-1] * e.shape[-2]) % 4 != 0:
                return done
            memo = self._head_memo(e, head)  # (None: the frame is not in the cache, or the byte cap is reached)
            if memo is None:
                continue
            if memo.get("k") is k and memo.get("stamp") == stamp and memo.get("term") is not None:
                done += 1
            else:
                todo.append((e, memo))
        for i0 in range(0, len(todo), batch):
            part = todo[i0:i0 + batch]
            x = torch.stack([e if e.dtype == torch.float32 else e.float() for e, _ in part])
            s1 = ops.dwconv7x7_bn_relu(x, w1[:cs], None if b1 is None else b1[:cs], scale=k["scale1"][:cs], shift=k["shift1"][:cs])
            terms = ops.conv1x1_mfma(s1, k["w2t_shared"], k["b2_zero"])  # [b, 256, h, w]
            for j, (_, memo) in enumerate(part):
                memo["k"], memo["term"], memo["stamp"] = k, terms[j:j + 1], stamp
            done += len(part)
        return done

    def prepare_clip(self, embeddings, batch=16):
        """Optional, for drivers that hold a clip's embeddings in one tensor (test.py:143-154 `embedding_memory`):
        prepares every frame of `embeddings` [F,C,h,w] up front, `batch` frames per launch, so the propagation loop
        finds each `embeddings[i]` ready.  Returns `embeddings` (in this model's storage type: pass the result on)."""
        if embeddings.dtype != self.emb_dtype:
            embeddings = embeddings.to(self.emb_dtype)
        if not embeddings.is_cuda or (torch.is_grad_enabled() and embeddings.requires_grad) or not self.cache_frames:
            return embeddings
        d = self._local_radius()
        if embeddings.shape[0] == 0 or self._frame_key(embeddings[0], d) is None:  # inference tensors: nothing to key on
            return embeddings
        # opting in to the clip-sized cache: room for this clip's frames (+ the default few), at most MAX_CACHED_FRAMES
        self._frame_cache_cap = min(MAX_CACHED_FRAMES, max(self._frame_cache_cap,
                                                           len(self._frame_cache) + embeddings.shape[0] + DEFAULT_CACHED_FRAMES))
        for i0 in range(0, embeddings.shape[0], batch):
            chunk = embeddings[i0:i0 + batch]
            for j, fr in enumerate(ops.prepare_frames(chunk, compute=self.compute, max_distance=d)):
                e = embeddings[i0 + j]
                fr.keep = e
                self._store_frame(self._frame_key(e, d), fr)
        self._trim_frames()
        return embeddings

    # ---- stored local-match volumes (r6): the label-independent half of IntVOS.py:345-434, once per frame pair -----------
    def prepare_local_volumes(self, embeddings, pairs="both", batch=32):
        """Optional, for drivers that hold a clip's embeddings (test.py:143-154 `embedding_memory`) and walk the clip more than
        once (every interaction round does: test.py:237-259, :276-295).  The window distances of the local match
        (local_pairwise_distances2, IntVOS.py:266-296) depend on the two embeddings only -- not on labels, objects or the round:
        they are computed HERE, `batch` frame pairs per launch, and kept; `prop_seghead` then runs only the label-dependent tail
        (IntVOS.py:398-432) for a pair it finds (ops.local_match_volume: 18 us instead of 49 at 480p, d = 12; the same bits).
        `embeddings`: [F, C, h, w] (or anything indexable by frame giving [C, h, w] views whose identity `prop_seghead` will see
        again); `pairs`: "both" = (t-1 -> t) and (t+1 -> t) for every t -- the forward and the backward propagation, "forward",
        "backward", or a list of (previous frame index, current frame index).  Kept until `invalidate_caches()`, LRU beyond
        `local_volume_cache_bytes`.  Returns the number of pairs now cached from this call."""
        d = self._local_radius()
        F_ = int(embeddings.shape[0])
        if d < 0 or F_ == 0 or not self.cache_frames or self.training or torch.cuda.is_current_stream_capturing():
            return 0
        if isinstance(pairs, str):
            fwd = [(t - 1, t) for t in range(1, F_)]
            bwd = [(t + 1, t) for t in range(F_ - 2, -1, -1)]
            pairs = {"both": fwd + bwd, "forward": fwd, "backward": bwd}[pairs]
        todo, hits, seen = [], 0, set()
        for (ip, ic) in pairs:
            ep, ec = embeddings[int(ip)], embeddings[int(ic)]
            if not ec.is_cuda or (torch.is_grad_enabled() and (ep.requires_grad or ec.requires_grad)):
                return 0
            kp, kc = self._frame_key(ep, d), self._frame_key(ec, d)
            if kp is None or kc is None:  # inference tensors: nothing to key on
                return 0
            if (kp, kc) in self._vol_cache:
                self._vol_cache.move_to_end((kp, kc))
                hits += 1
            elif (kp, kc) not in seen:
                seen.add((kp, kc))
                todo.append((kp, kc, ep, ec))
        if not todo:
            return hits
        h, w = todo[0][3].shape[-2:]
        per = ops.local_volume_bytes(int(h), int(w), d)
        todo = todo[:max(0, self.local_volume_cache_bytes // per)]  # what does not fit the cap stays on the fused kernel
        if not todo:
            return hits
        # ONE allocation for the call's volumes (a first-time device allocation of this size costs milliseconds: once, not per batch)
        store = torch.empty((len(todo), per // 4), dtype=torch.float32, device=todo[0][3].device)
        for i0 in range(0, len(todo), batch):
            part = todo[i0:i0 + batch]
            prevs = [self._prepared_frame(ep)[0] for (_, _, ep, _) in part]
            curs = [self._prepared_frame(ec)[0] for (_, _, _, ec) in part]
            vols = ops.local_volumes(prevs, curs, out=store[i0:i0 + len(part)])
            for j, (kp, kc, ep, ec) in enumerate(part):
                self._vol_store((kp, kc), vols[j], ep, ec)
        return hits + len(todo)

    def invalidate_local_volumes(self):
        self._vol_cache.clear()
        self._vol_cache_bytes = 0

    def local_volume_bytes_cached(self):
        return self._vol_cache_bytes

    def _vol_store(self, key, vol, ep, ec):
        # (the keys hold storage pointers: the embeddings stay alive with the entry)
        self._vol_cache[key] = [vol, ep, ec]
        self._vol_cache_bytes += vol.numel() * 4
        while self._vol_cache_bytes > self.local_volume_cache_bytes and len(self._vol_cache) > 1:
            _, old = self._vol_cache.popitem(last=False)
            self._vol_cache_bytes -= old[0].numel() * 4

    def _local_volume(self, prev_chw, cur_chw, fcur=None):
        """the stored volume of (previous frame, current frame), or None (then the fused kernel runs).  With
        `local_volume_lazy` a miss computes and keeps the pair's volume (one un-batched launch: 39 us at 480p, d = 12 -- slower
        than the fused kernel the first time, faster from the pair's second use on)."""
        if (not self._vol_cache and not self.local_volume_lazy) or torch.cuda.is_current_stream_capturing():
            return None
        d = self._local_radius()
        kp, kc = self._frame_key(prev_chw, d), self._frame_key(cur_chw, d)
        if kp is None or kc is None:
            return None
        hit = self._vol_cache.get((kp, kc))
        if hit is not None:
            self._vol_cache.move_to_end((kp, kc))
            return hit[0]
        if not self.local_volume_lazy or not self.cache_frames:
            return None
        fprev = self._prepared_frame(prev_chw)[0]
        fcur = fcur if fcur is not None else self._prepared_frame(cur_chw)[0]
        vol = ops.local_volumes([fprev], [fcur])[0]
        self._vol_store((kp, kc), vol, prev_chw, cur_chw)
        return vol

    def invalidate_caches(self):
        """Drop the prepared banks / frames and the heads' folded BatchNorm constants.  Called by train(),
        load_state_dict() and _apply() (device / dtype moves); call it yourself after writing to a parameter or an
        embedding through `.data` or a raw pointer -- such writes do not bump torch's version counters, which is all
        the identity keys can see."""
        self._bank_cache.clear()
        self._frame_cache.clear()
        self._memo_bytes = 0
        self.invalidate_local_volumes()
        self._dist_mirror.clear()
        self._frame_cache_cap = DEFAULT_CACHED_FRAMES
        for m in self.modules():
            if hasattr(m, "_fold_cache"):
                object.__setattr__(m, "_fold_cache", None)

    def train(self, mode=True):
        self.invalidate_caches()
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self.invalidate_caches()
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        if hasattr(self, "_bank_cache"):
            self.invalidate_caches()
        return super()._apply(fn, *args, **kwargs)

    # reference IntVOS.py:556-575
    def forward(self, x=None, ref_scribble_label=None, previous_frame_mask=None,
                normalize_nearest_neighbor_distances=True, use_local_map=True, seq_names=None, gt_ids=None,
                k_nearest_neighbors=1, global_map_tmp_dic=None, local_map_dics=None, interaction_num=None,
                start_annotated_frame=None, frame_num=None):
        x = self.extract_feature(x)
        ref_frame_embedding, previous_frame_embedding, current_frame_embedding = torch.split(
            x, split_size_or_sections=int(x.size(0) / 3), dim=0)
        if global_map_tmp_dic is None:
            return self.prop_seghead(ref_frame_embedding, previous_frame_embedding, current_frame_embedding,
                                     ref_scribble_label, previous_frame_mask,
                                     normalize_nearest_neighbor_distances, use_local_map, seq_names, gt_ids,
                                     k_nearest_neighbors, global_map_tmp_dic, local_map_dics, interaction_num,
                                     start_annotated_frame, frame_num, self.dynamic_seghead)
        dic, global_map_tmp_dic = self.prop_seghead(ref_frame_embedding, previous_frame_embedding,
                                                    current_frame_embedding, ref_scribble_label,
                                                    previous_frame_mask, normalize_nearest_neighbor_distances,
                                                    use_local_map, seq_names, gt_ids, k_nearest_neighbors,
                                                    global_map_tmp_dic, local_map_dics, interaction_num,
                                                    start_annotated_frame, frame_num, self.dynamic_seghead)
        return dic, global_map_tmp_dic

    # reference IntVOS.py:578-581
    def extract_feature(self, x, packed=False):
        """as the reference; the output is stored in this model's `emb_dtype`.  packed=True (inference): the per-frame
        operands of every frame of the batch are written right here, by one launch for the whole batch (prepare_clip):
        worthwhile when the caller keeps using THIS tensor's frames (a `torch.cat` of several batches copies them to new
        storage -- then call prepare_clip on the concatenated tensor instead)."""
        x = self.feature_extracter(x)
        if (packed and x.is_cuda and not self.training and not torch.is_grad_enabled() and self.cache_frames
                and x.dtype == torch.float32):
            # the embedding layer's epilogue fused with the frame prepare (ops.embed_finish): bn2 + relu2 + the storage cast +
            # the operands of every frame of the batch in ONE launch behind the framework's 1x1 GEMM -- the embedding is
            # written once and never re-read for packing (r3: three elementwise passes, then manet_frame_prepare's re-read)
            y = self.embedding_conv(self._separate_conv_bn_relu(x))
            scale, shift = ops.fold_bn(self.bn2)
            d = self._local_radius()
            emb, frames = ops.embed_finish(y, scale, shift, relu=True, emb_dtype=self.emb_dtype, compute=self.compute,
                                           max_distance=d)
            # (opting in to the clip-sized cache: room for what is cached already + this batch)
            self._frame_cache_cap = min(MAX_CACHED_FRAMES, max(self._frame_cache_cap,
                                                               len(self._frame_cache) + emb.shape[0] + DEFAULT_CACHED_FRAMES))
            for i, fr in enumerate(frames):
                e = emb[i]
                key = self._frame_key(e, d)
                if key is not None:
                    fr.keep = e
                    self._store_frame(key, fr)
            self._trim_frames()
            return emb
        x = self.semantic_embedding(x)
        if x.dtype != self.emb_dtype and not (torch.is_grad_enabled() and x.requires_grad):
            x = x.to(self.emb_dtype)
        if packed:
            x = self.prepare_clip(x, batch=max(1, x.shape[0]))
        return x

    def _separate_conv_bn_relu(self, x):
        """relu1(bn1(seperate_conv(x))) of the embedding head (IntVOS.py:537-539: depthwise 3x3 + BatchNorm + ReLU) in ONE launch
        of the head's depthwise kernel (r5): the 3x3 taps sit in the middle of a 7x7 kernel of zeros -- fmaf(x, 0, acc) = acc, the
        nine real taps in the order a 3x3 loop visits them, bn1 folded as in the head.  The framework runs the three modules as
        three launches (208 us per 480p frame against 40; a non-finite activation next to a zero tap becomes NaN here: Inf * 0).
        Falls back to the module sequence for anything else than a 3x3 / stride 1 / padding 1 depthwise layer on fp32 CUDA input."""
        conv, bn = self.seperate_conv, self.bn1
        ok = (x.is_cuda and x.dtype == torch.float32 and not self.training and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
              and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == conv.in_channels == conv.out_channels
              and conv.weight.dtype == torch.float32 and isinstance(bn, nn.BatchNorm2d) and bn.track_running_stats)
        if not ok:
            return self.relu1(bn(conv(x)))
        try:
            key = (conv.weight._version, conv.weight.data_ptr(), conv.weight.device)
        except RuntimeError:
            key = None
        hit = self.__dict__.get("_sep7")
        if hit is None or key is None or hit[0] != key:
            w7 = torch.zeros((conv.out_channels, 1, 7, 7), dtype=torch.float32, device=conv.weight.device)
            w7[:, :, 2:5, 2:5] = conv.weight.detach()
            hit = self.__dict__["_sep7"] = (key, w7)
        return ops.dwconv7x7_bn_relu(x.contiguous(), hit[1], conv.bias, bn=bn, relu=True)

    def prepare_bank(self, ref_frame_embedding, ref_scribble_label, seq_name, gt_id):
        """Sort / pack the annotated frame's memory bank NOW, on the current stream (this implementation only): what the first
        `prop_seghead` of an interaction round does on its way (`_prepared_bank`; the following frames then find it cached).
        A driver that issues the round's two directions on two HIP streams calls this before the streams fork -- otherwise
        the bank is built on one stream while the other may already match against it (the same goes for the heads' folded
        BatchNorm constants and the cached id vector: all made here).  `ref_frame_embedding` [1, C, h, w],
        `ref_scribble_label` [1, 1, h', w'] as passed to `prop_seghead`.  Returns the ops.PreparedBank."""
        _, _, h, w = ref_frame_embedding.shape
        if not ref_frame_embedding.is_cuda:
            raise RuntimeError("prepare_bank: embeddings must be on a HIP device (no CPU fallback)")
        if self.cfg.TEST_MODE:  # as prop_seghead (IntVOS.py:591-595)
            lab = ref_scribble_label.float()
        else:
            lab = F.interpolate(ref_scribble_label.float(), size=(h, w), mode="nearest")
        ref_lab = lab.int()[0].permute(1, 2, 0).reshape(-1)
        n_ids = _n_ids_from(gt_id, None)
        _obj_ids(n_ids, ref_frame_embedding.device)  # (the cached id vector, too, exists before any fork ...
        if not self.training and not torch.is_grad_enabled():  # ... and so do the heads' folded BatchNorm constants)
            for head in (self.dynamic_seghead, self.inter_seghead):
                if isinstance(head, DynamicSegHead):
                    head.layer1._folded(ref_frame_embedding.shape[1])
                    for layer in (head.layer2, head.layer3, head.layer4):
                        layer._folded()
        return self._prepared_bank(seq_name, ref_frame_embedding[0], ref_scribble_label[0],
                                   ref_frame_embedding[0].permute(1, 2, 0), ref_lab, n_ids)

    def global_maps(self, ref_frame_embedding, ref_scribble_label, embeddings, frame_nums, seq_name, gt_id,
                    stored_maps=None):
        """The label-INDEPENDENT half of `prop_seghead` for a block of frames (this implementation only): the global
        match of every frame of `embeddings` [f, C, h, w] against the annotated frame (`ref_frame_embedding` [1, C, h, w],
        `ref_scribble_label` [1, 1, h, w] at grid resolution -- TEST_MODE's convention, IntVOS.py:591-592), normalised
        (IntVOS.py:611-612) and min-merged (:615-622) with `stored_maps` [f, h*w*n_ids] (the frames' rows of
        `global_map_tmp_dic`; None = the first round's all-ones).  Returns float32 [f, h*w*n_ids]: what
        `prop_seghead(..., global_map_precomputed=...)` consumes.  Frame t's map needs only (bank, embedding_t, stored
        map_t): the frames of a clip shard over the GPUs of a node, the sequential chain (local match ->
        head -> mask) then runs on one.  Same kernels as inside prop_seghead: the same bits."""
        f, c, h, w = embeddings.shape
        n_ids = _n_ids_from(gt_id, None)
        if not embeddings.is_cuda:
            raise RuntimeError("global_maps: embeddings must be on a HIP device (no CPU fallback)")
        bank = self.prepare_bank(ref_frame_embedding, ref_scribble_label, seq_name, gt_id)
        out = torch.empty((f, h * w * n_ids), dtype=torch.float32, device=embeddings.device)
        if stored_maps is None:
            mem = torch.ones_like(out)
        else:
            mem = stored_maps.reshape(f, h * w * n_ids).to(torch.float32).clone()
        for i in range(f):
            fcur, _ = self._prepared_frame(embeddings[i])
            bank.match(fcur, normalize=True, mem=mem[i], out=out[i].view(h * w, n_ids))
        return out

    # reference IntVOS.py:583-681
    def prop_seghead(self, ref_frame_embedding=None, previous_frame_embedding=None, current_frame_embedding=None,
                     ref_scribble_label=None, previous_frame_mask=None, normalize_nearest_neighbor_distances=True,
                     use_local_map=True, seq_names=None, gt_ids=None, k_nearest_neighbors=1,
                     global_map_tmp_dic=None, local_map_dics=None, interaction_num=None,
                     start_annotated_frame=None, frame_num=None, dynamic_seghead=None, global_map_precomputed=None):
        """return: feature_embedding, global_match_map, local_match_map, previous_frame_mask

        global_map_precomputed (this implementation only; the reference's signature is a prefix of this one): dict
        seq_name -> {frame number -> float32 tensor of h*w*n_ids elements}: the normalised, min-merged global map of that
        frame as `global_maps()` returns it -- computed ahead of the sequential chain, e.g. by the other GPUs of the node
        (clip_parallel.propagate_round).  The global match of such a frame is skipped; the map is stored into
        `global_map_tmp_dic` exactly as the fused epilogue would have left it."""
        cfg = self.cfg
        dic_tmp = {}
        bs, c, h, w = current_frame_embedding.size()
        scaled_ref = []

        def scale_ref_scribble_label():
            # on demand: a propagation loop passes the same annotated frame for a whole clip and the prepared bank is a cache
            # hit -- the float -> int conversion of its labels (a launch per frame) is then never needed
            if not scaled_ref:
                if cfg.TEST_MODE:
                    lab = ref_scribble_label.float()
                else:
                    lab = F.interpolate(ref_scribble_label.float(), size=(h, w), mode="nearest")
                scaled_ref.append(lab.int())
            return scaled_ref[0]
        if (bs == 1 and previous_frame_mask.is_cuda and not previous_frame_mask.is_floating_point()
                and previous_frame_mask.numel() == previous_frame_mask.shape[-1] * previous_frame_mask.shape[-2]):
            # the same in one launch -- issued inside the loop below (ops.frame_begin), where it can carry the frame's two small
            # device writes along: the local map's pre-set and the distance weight (r5; each was a ~5 us launch of its own)
            scale_previous_frame_label = None
        else:
            scale_previous_frame_label = F.interpolate(previous_frame_mask.float(), size=(h, w), mode="nearest").int()
        for n in range(bs):
            # HWC views of the C-major embeddings: exactly what the kernels read coalesced
            seq_current_frame_embedding = current_frame_embedding[n].permute(1, 2, 0)
            seq_ref_frame_embedding = ref_frame_embedding[n].permute(1, 2, 0)
            seq_prev_frame_embedding = previous_frame_embedding[n].permute(1, 2, 0)
            def seq_ref_label_flat(n=n):
                return scale_ref_scribble_label()[n].permute(1, 2, 0).reshape(-1)
            n_ids = _n_ids_from(gt_ids[n], None)
            ref_obj_ids = _obj_ids(n_ids, current_frame_embedding.device)

            # ---- global map: match + normalise (:611-612) + min-merge with the memory (:615-622), fused
            mem = None
            if global_map_tmp_dic is not None:
                if seq_names[n] not in global_map_tmp_dic:
                    global_map_tmp_dic[seq_names[n]] = torch.ones(
                        (MAX_CLIP_FRAMES, h, w, n_ids, 1), dtype=torch.float32,
                        device=current_frame_embedding.device)
                mem = global_map_tmp_dic[seq_names[n]][frame_num[n]]  # contiguous slice, updated in place
            ref_emb, ref_lab = seq_ref_frame_embedding, seq_ref_label_flat  # (the labels: a callable until someone needs them)
            if k_nearest_neighbors > 1 and cfg.TEST_MODE:
                ref_lab = ref_lab()
                keep = ref_lab != -1
                ref_emb, ref_lab = ref_emb.reshape(-1, c)[keep], ref_lab[keep]
            bank, fcur, lpre, preset_done = None, None, None, False
            inference = current_frame_embedding.is_cuda and not (
                torch.is_grad_enabled() and (ref_frame_embedding.requires_grad or current_frame_embedding.requires_grad
                                             or previous_frame_embedding.requires_grad))
            fused_local = use_local_map and inference and self._local_radius() >= 0
            pre = None
            if global_map_precomputed is not None and seq_names[n] in global_map_precomputed:
                pre = global_map_precomputed[seq_names[n]].get(int(frame_num[n]))
            if pre is not None:
                if not inference or not normalize_nearest_neighbor_distances or mem is None:
                    raise ValueError("global_map_precomputed needs inference mode, normalised distances and a "
                                     "global_map_tmp_dic (it carries the merged map of global_maps())")
            elif k_nearest_neighbors == 1 and inference:
                bank = self._prepared_bank(seq_names[n], ref_frame_embedding[n], ref_scribble_label[n], ref_emb,
                                           ref_lab, n_ids)
            if inference and (bank is not None or fused_local):
                # this frame's operands: ONE read of its embedding (cache hit when the driver prepared the clip, or on
                # later interaction rounds); wide windows want the local map pre-set to 1.0 -- it rides in that launch
                lslot = None
                if fused_local and local_map_dics is not None:
                    # the local map is written straight into its slot of the local-map memory (IntVOS.py:645-647 stores it
                    # there anyway: one device copy per frame less); the tables are created as the reference creates them
                    local_map_tmp_dic, local_map_dist_dic = local_map_dics
                    dev_ = current_frame_embedding.device
                    if seq_names[n] not in local_map_dist_dic:
                        local_map_dist_dic[seq_names[n]] = torch.zeros(MAX_CLIP_FRAMES, MAX_INTERACTIONS, device=dev_)
                    if seq_names[n] not in local_map_tmp_dic:
                        local_map_tmp_dic[seq_names[n]] = torch.zeros((MAX_CLIP_FRAMES, MAX_INTERACTIONS, h, w, n_ids, 1),
                                                                     dtype=torch.float32, device=dev_)
                    tab_ = local_map_tmp_dic[seq_names[n]]
                    if (tab_.dtype == torch.float32 and tab_.is_contiguous()
                            and tuple(tab_.shape[2:]) == (h, w, n_ids, 1) and 0 < interaction_num <= tab_.shape[1]):
                        lslot = tab_[frame_num[n]][interaction_num - 1].view(h, w, n_ids)
                if lslot is not None:
                    lpre = lslot
                elif fused_local and self._local_radius() >= 11:
                    lpre = torch.empty((h, w, n_ids), dtype=torch.float32, device=current_frame_embedding.device)
                fcur, preset_done = self._prepared_frame(current_frame_embedding[n],
                                                         preset=lpre if self._local_radius() >= 11 else None)
            weight_written = False
            if scale_previous_frame_label is None:  # (bs == 1: the fast label resize, deferred to here)
                fill_t, w_dst, w_val = None, None, 0.0
                if (fused_local and lpre is not None and self._local_radius() >= 11 and not preset_done
                        and lpre.dtype == torch.float32 and lpre.is_contiguous()):
                    fill_t = lpre
                if fused_local and local_map_dics is not None and seq_names[n] in local_map_dics[1]:
                    tab_w = local_map_dics[1][seq_names[n]]
                    fn_, in_ = int(frame_num[n]), int(interaction_num) - 1
                    if (tab_w.dtype == torch.float32 and tab_w.is_cuda and tab_w.dim() == 2 and tab_w.is_contiguous()
                            and 0 <= fn_ < tab_w.shape[0] and 0 <= in_ < tab_w.shape[1] and fn_ != start_annotated_frame):
                        w_dst, w_val = tab_w[fn_][in_], 1.0 / abs(fn_ - start_annotated_frame)
                scale_previous_frame_label = ops.frame_begin(previous_frame_mask, (h, w), fill=fill_t, fill_value=1.0,
                                                             scalar_dst=w_dst, scalar_value=w_val)
                preset_done = preset_done or fill_t is not None
                weight_written = w_dst is not None
            if pre is not None:  # computed ahead of the chain (global_maps): out == mem after the fused min-merge
                mem.view(-1).copy_(pre.reshape(-1))
                nn_features_n = mem.view(1, h, w, n_ids, 1).clone()
            elif bank is not None:  # the propagation loop matches every frame against ONE annotated frame (test.py:237-259)
                nn_features_n = bank.match(fcur, normalize=bool(normalize_nearest_neighbor_distances),
                                           mem=mem).view(1, h, w, n_ids, 1)
            else:
                nn_features_n = ops.global_match(ref_emb, seq_current_frame_embedding, ref_lab() if callable(ref_lab) else ref_lab,
                                                 n_ids, k_nearest_neighbors=k_nearest_neighbors, compute=self.compute,
                                                 normalize=bool(normalize_nearest_neighbor_distances),
                                                 mem=mem).view(1, h, w, n_ids, 1)

            # ---- local map
            seq_previous_frame_label = scale_previous_frame_label[n].permute(1, 2, 0)
            if fused_local:
                vol = self._local_volume(previous_frame_embedding[n], current_frame_embedding[n], fcur)
                if vol is not None:
                    # the pair's window distances are stored (prepare_local_volumes): only the label-dependent tail runs
                    prev_frame_nn_features_n = ops.local_match_volume(
                        vol, fcur, seq_previous_frame_label, n_ids, out=lpre,
                        out_is_preset=preset_done).view(1, h, w, n_ids, 1)
                else:
                    # the previous frame was the current frame of the last step (test.py:259): its plane is cached
                    fprev, _ = self._prepared_frame(previous_frame_embedding[n])
                    prev_frame_nn_features_n = ops.local_match_frames(
                        fprev, fcur, seq_previous_frame_label, n_ids, out=lpre,
                        out_is_preset=preset_done).view(1, h, w, n_ids, 1)
            elif use_local_map:
                prev_frame_nn_features_n = local_previous_frame_nearest_neighbor_features_per_object(
                    prev_frame_embedding=seq_prev_frame_embedding, query_embedding=seq_current_frame_embedding,
                    prev_frame_labels=seq_previous_frame_label, gt_ids=ref_obj_ids,
                    max_distance=cfg.MODEL_MAX_LOCAL_DISTANCE)
            else:
                prev_frame_nn_features_n = ops.global_match(
                    seq_prev_frame_embedding, seq_current_frame_embedding, seq_previous_frame_label.reshape(-1),
                    n_ids, k_nearest_neighbors=k_nearest_neighbors, compute=self.compute,
                    normalize=True).view(1, h, w, n_ids, 1)

            # ---- local map memory (:638-661)
            if local_map_dics is not None:
                local_map_tmp_dic, local_map_dist_dic = local_map_dics
                if seq_names[n] not in local_map_dist_dic:
                    local_map_dist_dic[seq_names[n]] = torch.zeros(MAX_CLIP_FRAMES, MAX_INTERACTIONS,
                                                                   device=prev_frame_nn_features_n.device)
                if seq_names[n] not in local_map_tmp_dic:
                    local_map_tmp_dic[seq_names[n]] = torch.zeros_like(prev_frame_nn_features_n).unsqueeze(
                        0).repeat(MAX_CLIP_FRAMES, MAX_INTERACTIONS, 1, 1, 1, 1)
                dist_tab, map_tab = local_map_dist_dic[seq_names[n]], local_map_tmp_dic[seq_names[n]]
                # python arithmetic first, as the reference: frame == annotated frame raises ZeroDivisionError
                weight = 1.0 / (abs(frame_num[n] - start_annotated_frame))
                # (host mirror of the weights this module wrote into THIS table: the comparison a few lines down then needs no
                # device read -- in the reference it is a device-to-host synchronisation per frame from the second round on)
                mirror = self._mirror_of(seq_names[n], dist_tab)  # (validated BEFORE this call's own write)
                # (fill_, not `tab[i][j] = weight`: indexed assignment of a python number to a device tensor goes through a
                # one-element HOST tensor and a blocking host-to-device copy -- the host then waits for the whole frame's queue,
                # ~0.85 ms, and the device idles ~70 us per frame until the next launch arrives; fill_ is one async launch)
                if not weight_written:  # (else: it rode in ops.frame_begin's launch -- a raw write, the version counter stands)
                    dist_tab[frame_num[n]][interaction_num - 1].fill_(weight)
                fkey = int(frame_num[n])
                if mirror is not None:
                    mirror[1] = dist_tab._version
                    # (as the table holds it: rounded to float32)
                    mirror[2][(fkey, interaction_num - 1)] = float(np.float32(weight)) if dist_tab.dtype == torch.float32 else None
                slot_ = map_tab[frame_num[n]][interaction_num - 1]
                if prev_frame_nn_features_n.data_ptr() != slot_.data_ptr():  # (not written in place above)
                    slot_.copy_(prev_frame_nn_features_n.squeeze(0).detach())
                newer_wins = True  # (first round: there is nothing older)
                if interaction_num > 1:
                    now_ = before_ = None
                    if mirror is not None:
                        now_, before_ = mirror[2][(fkey, interaction_num - 1)], mirror[2].get((fkey, interaction_num - 2))
                    if now_ is not None and before_ is not None:
                        newer_wins = now_ > before_
                    else:  # a table this module did not fill (or not in this process): ask the device, as the reference does
                        newer_wins = bool(dist_tab[frame_num[n]][interaction_num - 1] > dist_tab[frame_num[n]][interaction_num - 2])
                prev_frame_nn_features_n = map_tab[frame_num[n]][interaction_num - (1 if newer_wins else 2)].unsqueeze(0)
                local_map_dics = (local_map_tmp_dic, local_map_dist_dic)

            # ---- head input [n_ids, C+3, h, w] (:663-673)
            pred_ = None
            native_maps = (inference and nn_features_n.is_cuda and nn_features_n.dtype == torch.float32
                           and prev_frame_nn_features_n.dtype == torch.float32
                           and nn_features_n.numel() == h * w * n_ids and prev_frame_nn_features_n.numel() == h * w * n_ids)
            if (native_maps and isinstance(dynamic_seghead, DynamicSegHead) and not dynamic_seghead.training
                    and not torch.is_grad_enabled()  # (eval() with grad mode on: the literal, differentiable head below -- ADVICE r5)
                    and dynamic_seghead.layer1.conv1.kernel_size == (7, 7) and (h * w) % 4 == 0):
                # layer 1 straight from the two maps and the labels (r5): input assembly + per-object depthwise + its 1x1 + the
                # (memoised) shared half + ReLU in one launch; layers 2-4 as ever
                emb_f = current_frame_embedding[n]
                emb_f = emb_f if emb_f.dtype == torch.float32 else emb_f.float()
                x1 = _layer1_fused(dynamic_seghead.layer1, emb_f.unsqueeze(0), nn_features_n, prev_frame_nn_features_n,
                                   seq_previous_frame_label, n_ids, (h, w),
                                   memo=self._head_memo(current_frame_embedding[n], dynamic_seghead))
                if x1 is not None:
                    pred_ = dynamic_seghead._tail(x1)
            if pred_ is not None:
                pass
            elif native_maps:
                # the three per-object channels written by one launch (no arange / compare / permute / cat kernels)
                per_object = ops.head_inputs(nn_features_n, prev_frame_nn_features_n, seq_previous_frame_label, n_ids, (h, w))
            else:
                to_cat_previous_frame = (seq_previous_frame_label.float() == ref_obj_ids.float())
                to_cat_nn_feature_n = nn_features_n.squeeze(0).permute(2, 3, 0, 1)
                to_cat_previous_frame = to_cat_previous_frame.unsqueeze(-1).permute(2, 3, 0, 1).float()
                to_cat_prev_frame_nn_feature_n = prev_frame_nn_features_n.squeeze(0).permute(2, 3, 0, 1)
                per_object = torch.cat((to_cat_nn_feature_n, to_cat_prev_frame_nn_feature_n, to_cat_previous_frame), 1)
            if pred_ is None:
                pred_ = _run_head(dynamic_seghead, current_frame_embedding[n], per_object,
                                  memo=self._head_memo(current_frame_embedding[n], dynamic_seghead) if inference else None)
            dic_tmp[seq_names[n]] = pred_.permute(1, 0, 2, 3)

        if global_map_tmp_dic is None:
            return dic_tmp
        if local_map_dics is None:
            return dic_tmp, global_map_tmp_dic
        return dic_tmp, global_map_tmp_dic, local_map_dics

    # reference IntVOS.py:683-764
    def int_seghead(self, ref_frame_embedding=None, ref_scribble_label=None, prev_round_label=None,
                    normalize_nearest_neighbor_distances=True, global_map_tmp_dic=None, local_map_dics=None,
                    interaction_num=None, seq_names=None, gt_ids=None, k_nearest_neighbors=1, frame_num=None,
                    first_inter=True):
        cfg = self.cfg
        dic_tmp = {}
        bs, c, h, w = ref_frame_embedding.size()
        scale_ref_scribble_label = F.interpolate(ref_scribble_label.float(), size=(h, w), mode="nearest").int()
        if not first_inter:
            scale_prev_round_label = F.interpolate(prev_round_label.float(), size=(h, w), mode="nearest").int()
        for n in range(bs):
            n_ids = _n_ids_from(gt_ids[n], None)
            gt_id = torch.arange(0, n_ids, dtype=torch.int32, device=ref_frame_embedding.device)
            seq_ref_frame_embedding = ref_frame_embedding[n].permute(1, 2, 0)
            seq_ref_scribble_label = scale_ref_scribble_label[n].permute(1, 2, 0)
            # ---- local map of the annotated frame against itself (:709-711)
            if (ref_frame_embedding.is_cuda and self._local_radius() >= 0
                    and not (torch.is_grad_enabled() and ref_frame_embedding.requires_grad)):
                fref, _ = self._prepared_frame(ref_frame_embedding[n])
                vol = self._local_volume(ref_frame_embedding[n], ref_frame_embedding[n], fref)
                if vol is not None:
                    nn_features_n = ops.local_match_volume(vol, fref, seq_ref_scribble_label, n_ids).view(1, h, w, n_ids, 1)
                else:
                    nn_features_n = ops.local_match_frames(fref, fref, seq_ref_scribble_label, n_ids).view(1, h, w, n_ids, 1)
            else:
                nn_features_n = local_previous_frame_nearest_neighbor_features_per_object(
                    prev_frame_embedding=seq_ref_frame_embedding, query_embedding=seq_ref_frame_embedding,
                    prev_frame_labels=seq_ref_scribble_label, gt_ids=gt_id, max_distance=cfg.MODEL_MAX_LOCAL_DISTANCE)
            # ---- global map update (:716-723): min-merge of THIS map into the stored one
            if seq_names[n] not in global_map_tmp_dic:
                global_map_tmp_dic[seq_names[n]] = torch.ones_like(nn_features_n).repeat(MAX_CLIP_FRAMES, 1, 1, 1, 1)
            # (the reference stores the merged map detached and never feeds it to the head: IntVOS.py:718-723)
            merged = nn_features_n.detach().clone()
            ops.normalize_merge_(merged, global_map_tmp_dic[seq_names[n]][frame_num[n]], normalize=False)
            # ---- local map memory (:725-736)
            if local_map_dics is not None:
                local_map_tmp_dic, local_map_dist_dic = local_map_dics
                if seq_names[n] not in local_map_dist_dic:
                    local_map_dist_dic[seq_names[n]] = torch.zeros(MAX_CLIP_FRAMES, MAX_INTERACTIONS,
                                                                   device=nn_features_n.device)
                if seq_names[n] not in local_map_tmp_dic:
                    local_map_tmp_dic[seq_names[n]] = torch.ones_like(nn_features_n).unsqueeze(0).repeat(
                        MAX_CLIP_FRAMES, MAX_INTERACTIONS, 1, 1, 1, 1)
                dist_tab = local_map_dist_dic[seq_names[n]]
                mirror = self._mirror_of(seq_names[n], dist_tab)
                dist_tab[frame_num[n]][interaction_num - 1].fill_(0)  # (async; see prop_seghead)
                if mirror is not None:  # (this write is the module's own: the mirror follows it)
                    mirror[1] = dist_tab._version
                    mirror[2][(int(frame_num[n]), interaction_num - 1)] = 0.0
                local_map_dics = (local_map_tmp_dic, local_map_dist_dic)
            # ---- head input (:741-760)
            to_cat_scribble_mask_to_cat = (seq_ref_scribble_label.float() == gt_id.float())
            to_cat_scribble_mask_to_cat = to_cat_scribble_mask_to_cat.unsqueeze(-1).permute(2, 3, 0, 1).float()
            if not first_inter:
                seq_prev_round_label = scale_prev_round_label[n].permute(1, 2, 0)
                to_cat_prev_round_to_cat = (seq_prev_round_label.float() == gt_id.float())
                to_cat_prev_round_to_cat = to_cat_prev_round_to_cat.unsqueeze(-1).permute(2, 3, 0, 1).float()
            else:
                to_cat_prev_round_to_cat = torch.zeros_like(to_cat_scribble_mask_to_cat)
                to_cat_prev_round_to_cat[0] = 1.0
            per_object = torch.cat((to_cat_scribble_mask_to_cat, to_cat_prev_round_to_cat), 1)
            pred_ = _run_head(self.inter_seghead, ref_frame_embedding[n], per_object,
                              memo=self._head_memo(ref_frame_embedding[n], self.inter_seghead)
                              if (ref_frame_embedding.is_cuda and not torch.is_grad_enabled()) else None)
            dic_tmp[seq_names[n]] = pred_.permute(1, 0, 2, 3)
        if local_map_dics is None:
            return dic_tmp
        return dic_tmp, local_map_dics
