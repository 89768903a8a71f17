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

# arithmetic of the QK^T contraction: "f32" (exact fp32 MFMA) | "bf16" | "bf16x3"
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
        key = (cs,) + tuple((t.data_ptr(), t._version) if t is not None else None for t in src)
        hit = getattr(self, "_fold_cache", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        with torch.no_grad():
            scale1, shift1 = ops.fold_bn(self.bn1)
            scale2, shift2 = ops.fold_bn(self.bn2)
            w2 = (self.conv2.weight.detach().float() * scale2[:, None, None, None]).contiguous()
            b2 = (self.conv2.bias.detach().float() * scale2 + shift2 if self.conv2.bias is not None else shift2).contiguous()
            val = {"scale1": scale1.contiguous(), "shift1": shift1.contiguous(), "w2": w2, "b2": b2,
                   "w2_shared": w2[:, :cs].contiguous(), "w2_object": w2[:, cs:].contiguous()}
        object.__setattr__(self, "_fold_cache", (key, val))  # plain attribute: not a buffer, not in the state dict
        return val

    def forward(self, x, relu_in=False, defer_relu=False):
        """relu_in / defer_relu (inference fast path only, used by DynamicSegHead): the block's last ReLU is left to the
        NEXT block, whose fused depthwise kernel reads its input through max(x, 0) -- one elementwise pass over the
        activation less per block, same values."""
        if self._fast(x):
            k = self._folded()
            x = ops.dwconv7x7_bn_relu(x, self.conv1.weight, self.conv1.bias, scale=k["scale1"], shift=k["shift1"],
                                      relu_in=relu_in)
            y = F.conv2d(x, k["w2"], k["b2"])
            return y if defer_relu else y.relu_()
        assert not relu_in and not defer_relu
        x = self.relu1(self.bn1(self.conv1(x)))
        x = self.relu2(self.bn2(self.conv2(x)))
        return x

    def forward_shared(self, shared, per_object, defer_relu=False):
        """The block applied to cat([shared repeated n times, per_object], 1) without building that tensor
        (IntVOS.py:665-670 repeats the C-channel embedding once per object): the depthwise stage and the
        1x1 stage are linear in the channel groups, so the shared group is processed once.
        shared [1, Cs, h, w], per_object [n, Cp, h, w], Cs + Cp == in_dim."""
        cs = shared.shape[1]
        k = self._folded(cs)
        scale, shift = k["scale1"], k["shift1"]
        w1, b1 = self.conv1.weight, self.conv1.bias
        s1 = ops.dwconv7x7_bn_relu(shared, w1[:cs], b1[:cs], scale=scale[:cs], shift=shift[:cs])
        p1 = ops.dwconv7x7_bn_relu(per_object, w1[cs:], b1[cs:], scale=scale[cs:], shift=shift[cs:])
        y = F.conv2d(p1, k["w2_object"], k["b2"])
        y += F.conv2d(s1, k["w2_shared"])  # broadcast over the objects
        return y if defer_relu else y.relu_()


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
        """layers 2-4 + the 1x1 output conv on the inference fast path; x = layer1's output BEFORE its last ReLU
        (each block's ReLU is applied by the next block's fused depthwise kernel as it reads)"""
        x = self.layer2(x, relu_in=True, defer_relu=True)
        x = self.layer3(x, relu_in=True, defer_relu=True)
        if self.conv.kernel_size == (1, 1) and self.conv.out_channels == 1:
            # output layer fused with layer4's ReLU: one pass over the activation (ops.relu_conv1x1_c1)
            return ops.relu_conv1x1_c1(self.layer4(x, relu_in=True, defer_relu=True), self.conv.weight, self.conv.bias)
        return self.conv(self.layer4(x, relu_in=True))

    def forward(self, x):
        if self.layer1._fast(x):
            return self._tail(self.layer1(x, defer_relu=True))
        return self.conv(self.layer4(self.layer3(self.layer2(self.layer1(x)))))

    def forward_shared(self, shared, per_object):
        """forward(cat([shared.repeat(n,1,1,1), per_object], 1)) without materialising the input"""
        return self._tail(self.layer1.forward_shared(shared, per_object, defer_relu=True))


def _run_head(head, embedding_chw, per_object):
    """head(cat([embedding repeated per object, per_object], 1)) (IntVOS.py:665-671, :741-758).  In
    inference on the GPU a DynamicSegHead takes the shared-embedding route (no repeat / cat of the C-channel
    embedding, depthwise stage of those channels computed once); otherwise the reference's literal form."""
    n = per_object.shape[0]
    if (isinstance(head, DynamicSegHead) and not head.training and not torch.is_grad_enabled()
            and embedding_chw.is_cuda and embedding_chw.dtype == torch.float32):
        return head.forward_shared(embedding_chw.unsqueeze(0), per_object)
    return head(torch.cat((embedding_chw.unsqueeze(0).repeat((n, 1, 1, 1)), per_object), 1))


# --------------------------------------------------------------------------------------------------
MAX_CLIP_FRAMES = 104       # hard-coded clip length of the reference's memories (IntVOS.py:617,645)
MAX_INTERACTIONS = 9        # IntVOS.py:641,645


class IntVOS(nn.Module):
    """reference IntVOS.py:530-764: same constructor, methods, dict conventions, state-dict keys."""

    def __init__(self, cfg, feature_extracter):
        super().__init__()
        set_cfg(cfg)
        self.cfg = cfg
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
        self._bank_cache = {}  # seq_name -> (identity key, ops.PreparedBank, keyed tensors): see _prepared_bank
        self.dynamic_seghead = DynamicSegHead()  # propagation head
        if cfg.MODEL_USEIntSeg:
            self.inter_seghead = IntSegHead(in_dim=cfg.MODEL_SEMANTIC_EMBEDDING_DIM + 3)
        else:
            self.inter_seghead = DynamicSegHead(in_dim=cfg.MODEL_SEMANTIC_EMBEDDING_DIM + 2)  # interaction head

    def _prepared_bank(self, seq_name, ref_emb_chw, ref_label, ref_emb_hwc, ref_lab_flat, n_ids):
        """The sorted / packed memory bank of the annotated frame, reused while the caller keeps passing the SAME
        embedding and scribble tensors (identity = storage pointer, shape, strides and torch's in-place version
        counter): test.py:237-259 / :276-295 propagate a whole clip against one annotated frame, so the bank is
        sorted and packed once per interaction instead of once per frame.  One bank per sequence name."""
        key = (ref_emb_chw.data_ptr(), tuple(ref_emb_chw.shape), tuple(ref_emb_chw.stride()), ref_emb_chw._version,
               ref_emb_chw.dtype, ref_label.data_ptr(), tuple(ref_label.shape), ref_label._version, ref_label.dtype,
               n_ids, COMPUTE, bool(self.cfg.TEST_MODE))
        hit = self._bank_cache.get(seq_name)
        if hit is not None and hit[0] == key:
            return hit[1]
        bank = ops.PreparedBank(ref_emb_hwc, ref_lab_flat, n_ids, compute=COMPUTE)
        # keep the keyed tensors alive so that their storage pointers cannot be recycled under the key
        self._bank_cache[seq_name] = (key, bank, ref_emb_chw, ref_label)
        return bank

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
    def extract_feature(self, x):
        x = self.feature_extracter(x)
        x = self.semantic_embedding(x)
        return x

    # reference IntVOS.py:583-681
    def prop_seghead(self, ref_frame_embedding=None, previous_frame_embedding=None, current_frame_embedding=None,
                     ref_scribble_label=None, previous_frame_mask=None, normalize_nearest_neighbor_distances=True,
                     use_local_map=True, seq_names=None, gt_ids=None, k_nearest_neighbors=1,
                     global_map_tmp_dic=None, local_map_dics=None, interaction_num=None,
                     start_annotated_frame=None, frame_num=None, dynamic_seghead=None):
        """return: feature_embedding, global_match_map, local_match_map, previous_frame_mask"""
        cfg = self.cfg
        dic_tmp = {}
        bs, c, h, w = current_frame_embedding.size()
        if cfg.TEST_MODE:
            scale_ref_scribble_label = ref_scribble_label.float()
        else:
            scale_ref_scribble_label = F.interpolate(ref_scribble_label.float(), size=(h, w), mode="nearest")
        scale_ref_scribble_label = scale_ref_scribble_label.int()
        scale_previous_frame_label = F.interpolate(previous_frame_mask.float(), size=(h, w), mode="nearest").int()
        for n in range(bs):
            # HWC views of the C-major embeddings: exactly what the kernels read coalesced
            seq_current_frame_embedding = current_frame_embedding[n].permute(1, 2, 0)
            seq_ref_frame_embedding = ref_frame_embedding[n].permute(1, 2, 0)
            seq_prev_frame_embedding = previous_frame_embedding[n].permute(1, 2, 0)
            seq_ref_scribble_label = scale_ref_scribble_label[n].permute(1, 2, 0)
            n_ids = _n_ids_from(gt_ids[n], None)
            ref_obj_ids = torch.arange(0, n_ids, dtype=torch.int32, device=current_frame_embedding.device)

            # ---- global map: match + normalise (:611-612) + min-merge with the memory (:615-622), fused
            mem = None
            if global_map_tmp_dic is not None:
                if seq_names[n] not in global_map_tmp_dic:
                    global_map_tmp_dic[seq_names[n]] = torch.ones(
                        (MAX_CLIP_FRAMES, h, w, n_ids, 1), dtype=torch.float32,
                        device=current_frame_embedding.device)
                mem = global_map_tmp_dic[seq_names[n]][frame_num[n]]  # contiguous slice, updated in place
            ref_emb, ref_lab = seq_ref_frame_embedding, seq_ref_scribble_label.reshape(-1)
            if k_nearest_neighbors > 1 and cfg.TEST_MODE:
                keep = ref_lab != -1
                ref_emb, ref_lab = ref_emb.reshape(-1, c)[keep], ref_lab[keep]
            bank = None
            if (k_nearest_neighbors == 1 and current_frame_embedding.is_cuda
                    and not (torch.is_grad_enabled() and (ref_frame_embedding.requires_grad
                                                          or current_frame_embedding.requires_grad))):
                bank = self._prepared_bank(seq_names[n], ref_frame_embedding[n], ref_scribble_label[n], ref_emb,
                                           ref_lab, n_ids)
            if bank is not None:  # the propagation loop matches every frame against ONE annotated frame (test.py:237-259)
                nn_features_n = bank.match(seq_current_frame_embedding,
                                           normalize=bool(normalize_nearest_neighbor_distances),
                                           mem=mem).view(1, h, w, n_ids, 1)
            else:
                nn_features_n = ops.global_match(ref_emb, seq_current_frame_embedding, ref_lab, n_ids,
                                                 k_nearest_neighbors=k_nearest_neighbors, compute=COMPUTE,
                                                 normalize=bool(normalize_nearest_neighbor_distances),
                                                 mem=mem).view(1, h, w, n_ids, 1)

            # ---- local map
            seq_previous_frame_label = scale_previous_frame_label[n].permute(1, 2, 0)
            if use_local_map:
                prev_frame_nn_features_n = local_previous_frame_nearest_neighbor_features_per_object(
                    prev_frame_embedding=seq_prev_frame_embedding, query_embedding=seq_current_frame_embedding,
                    prev_frame_labels=seq_previous_frame_label, gt_ids=ref_obj_ids,
                    max_distance=cfg.MODEL_MAX_LOCAL_DISTANCE)
            else:
                prev_frame_nn_features_n = ops.global_match(
                    seq_prev_frame_embedding, seq_current_frame_embedding, seq_previous_frame_label.reshape(-1),
                    n_ids, k_nearest_neighbors=k_nearest_neighbors, compute=COMPUTE,
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
                dist_tab[frame_num[n]][interaction_num - 1] = weight
                map_tab[frame_num[n]][interaction_num - 1] = prev_frame_nn_features_n.squeeze(0).detach()
                if interaction_num == 1:
                    prev_frame_nn_features_n = map_tab[frame_num[n]][interaction_num - 1].unsqueeze(0)
                elif dist_tab[frame_num[n]][interaction_num - 1] > dist_tab[frame_num[n]][interaction_num - 2]:
                    prev_frame_nn_features_n = map_tab[frame_num[n]][interaction_num - 1].unsqueeze(0)
                else:
                    prev_frame_nn_features_n = map_tab[frame_num[n]][interaction_num - 2].unsqueeze(0)
                local_map_dics = (local_map_tmp_dic, local_map_dist_dic)

            # ---- head input [n_ids, C+3, h, w] (:663-673)
            to_cat_previous_frame = (seq_previous_frame_label.float() == ref_obj_ids.float())
            to_cat_nn_feature_n = nn_features_n.squeeze(0).permute(2, 3, 0, 1)
            to_cat_previous_frame = to_cat_previous_frame.unsqueeze(-1).permute(2, 3, 0, 1).float()
            to_cat_prev_frame_nn_feature_n = prev_frame_nn_features_n.squeeze(0).permute(2, 3, 0, 1)
            per_object = torch.cat((to_cat_nn_feature_n, to_cat_prev_frame_nn_feature_n, to_cat_previous_frame), 1)
            pred_ = _run_head(dynamic_seghead, current_frame_embedding[n], per_object)
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
                local_map_dist_dic[seq_names[n]][frame_num[n]][interaction_num - 1] = 0
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
            pred_ = _run_head(self.inter_seghead, ref_frame_embedding[n], per_object)
            dic_tmp[seq_names[n]] = pred_.permute(1, 0, 2, 3)
        if local_map_dics is None:
            return dic_tmp
        return dic_tmp, local_map_dics
