"""torch.autograd.Function wrappers of the matching path's training entry points.

The reference's matching functions are plain PyTorch, so autograd differentiates them for free and
train_stage1.py:126-156 / train_stage2.py back-propagate through IntVOS.forward.  The HIP path takes raw
pointers, so the backward is explicit:

  GlobalMatchFn   nearest_neighbor_features_per_object, k = 1 (reference IntVOS.py:160-210): torch.min sends the
                  gradient to ONE bank row per (query pixel, object) -- the forward kernel records it
                  (manet_global_match_arg_f32), the backward is a gather / scatter-add
                  (manet_global_match_backward_f32).
  GlobalMatchTopkFn  the same with k > 1 (IntVOS.py:87-94): k exact arg-min passes, a gather / scatter-add per rank.
  LocalMatchFn    local_previous_frame_nearest_neighbor_features_per_object, downsample on (:345-434): the
                  forward records the winning window offset and keeps the normalised pooled volume; the
                  backward walks min -> where -> bilinear -> sigmoid -> (x - y)^2 -> avg_pool2d in reverse.
  LocalMatchFullFn  the same with MODEL_LOCAL_DOWNSAMPLE = False (:299-313): raw full-resolution distances, no sigmoid / bilinear.
  CorrelationFn   correlation_package (correlation.py:7-45): forward + manet_correlation_backward_f32.

`ops.global_match` / `ops.local_match` / `ops.correlation_forward` route here when grad mode is on and an
embedding requires grad; normalisation and the min-merge with the stored map stay ordinary torch ops on the
result (they are element-wise and torch differentiates them).  fp32 only; anything else raises.
"""
import ctypes

import torch

from . import _lib


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def _scratch(device, nbytes):
    return torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)


class GlobalMatchFn(torch.autograd.Function):
    """out [N, n_ids] raw distances; differentiable w.r.t. reference [M0, C] and query [N, C] (any strides)."""

    @staticmethod
    def forward(ctx, ref, qry, labels, n_ids):
        lib = _lib.load()
        M0, C = ref.shape
        N = qry.shape[0]
        dev = qry.device
        nbytes = ctypes.c_size_t(0)
        _lib.check(lib.manet_global_match_arg_workspace_bytes(N, M0, C, n_ids, ctypes.byref(nbytes)),
                   "manet_global_match_arg_workspace_bytes")
        ws = _scratch(dev, nbytes.value)
        out = torch.empty((N, n_ids), dtype=torch.float32, device=dev)
        arg = torch.empty((N, n_ids), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.manet_global_match_arg_f32(qry.data_ptr(), qry.stride(0), qry.stride(1), ref.data_ptr(),
                                                ref.stride(0) if M0 > 0 else C, ref.stride(1) if M0 > 0 else 1,
                                                labels.data_ptr(), N, M0, C, n_ids, out.data_ptr(), arg.data_ptr(),
                                                ws.data_ptr(), ws.numel(), _stream_ptr(dev))
        _lib.check(rc, "manet_global_match_arg_f32")
        ctx.save_for_backward(ref, qry, arg)
        ctx.n_ids = n_ids
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, grad_out, _grad_arg):
        lib = _lib.load()
        ref, qry, arg = ctx.saved_tensors
        M0, C = ref.shape
        N = qry.shape[0]
        dev = qry.device
        g = grad_out.contiguous().float()
        need_ref, need_qry = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_ref or need_qry):
            return None, None, None, None
        # gradients in the inputs' own memory order (C-major embeddings stay C-major); a gradient nobody asked for (the
        # frozen reference frame of fine-tuning) is neither zero-filled nor scattered into
        gq = gr = None
        if need_qry:
            gq = torch.empty_strided(qry.shape, qry.stride(), dtype=torch.float32, device=dev) \
                if _dense(qry) else torch.empty(qry.shape, dtype=torch.float32, device=dev)
        if need_ref:
            gr = torch.empty_strided(ref.shape, ref.stride(), dtype=torch.float32, device=dev) \
                if _dense(ref) else torch.empty(ref.shape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.manet_global_match_backward_f32(qry.data_ptr(), qry.stride(0), qry.stride(1), ref.data_ptr(),
                                                     ref.stride(0) if M0 > 0 else C, ref.stride(1) if M0 > 0 else 1,
                                                     arg.data_ptr(), g.data_ptr(), N, M0, C, ctx.n_ids,
                                                     None if gq is None else gq.data_ptr(),
                                                     C if gq is None else gq.stride(0), 1 if gq is None else gq.stride(1),
                                                     None if gr is None else gr.data_ptr(),
                                                     C if (gr is None or M0 == 0) else gr.stride(0),
                                                     1 if (gr is None or M0 == 0) else gr.stride(1), _stream_ptr(dev))
        _lib.check(rc, "manet_global_match_backward_f32")
        return gr, gq, None, None


class GlobalMatchTopkFn(torch.autograd.Function):
    """nearest_neighbor_features_per_object with k_nearest_neighbors > 1 (reference IntVOS.py:87-94): out [N, n_ids] =
    mean of the k smallest distances per (query, object), entries past the object's row count replaced by the farthest real
    neighbour.  The reference gets the gradient from autograd through topk -> where -> max -> mean: each real neighbour of rank
    j receives g / k, and the farthest real one additionally the share of every replaced entry ((k - v) g / k with v real
    neighbours; nothing at all when the object has no row: `dists * valid` multiplies the padding by zero).  Forward:
    manet_global_match_topk_arg_f32 (k exact passes of the arg-min kernel); backward: one gather / scatter-add launch per rank."""

    @staticmethod
    def forward(ctx, ref, qry, labels, n_ids, k):
        lib = _lib.load()
        M0, C = ref.shape
        N = qry.shape[0]
        dev = qry.device
        nbytes = ctypes.c_size_t(0)
        _lib.check(lib.manet_global_match_topk_arg_workspace_bytes(N, M0, C, n_ids, ctypes.byref(nbytes)),
                   "manet_global_match_topk_arg_workspace_bytes")
        ws = _scratch(dev, nbytes.value)
        d = torch.empty((k, N, n_ids), dtype=torch.float32, device=dev)
        arg = torch.empty((k, N, n_ids), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.manet_global_match_topk_arg_f32(qry.data_ptr(), qry.stride(0), qry.stride(1), ref.data_ptr(),
                                                     ref.stride(0) if M0 > 0 else C, ref.stride(1) if M0 > 0 else 1,
                                                     labels.data_ptr(), N, M0, C, n_ids, k, d.data_ptr(), arg.data_ptr(),
                                                     ws.data_ptr(), ws.numel(), _stream_ptr(dev))
        _lib.check(rc, "manet_global_match_topk_arg_f32")
        # IntVOS.py:88-94 on the k sorted distances (rank along dim 0)
        valid = d < 1e20
        masked = d * valid.float()
        pad, jstar = masked.max(dim=0, keepdim=True)
        out = torch.where(valid, d, pad.expand_as(d)).mean(dim=0)
        # per-rank weights of the incoming gradient: 1/k for a real neighbour, + (k - v)/k on the rank the padding came from
        # (times that rank's own validity: with no real neighbour the padding is 0 * d -- no gradient)
        nvalid = valid.sum(dim=0, keepdim=True)
        wgt = valid.float() / k
        extra = (k - nvalid).float() / k
        wgt.scatter_add_(0, jstar, extra * torch.gather(valid.float(), 0, jstar))
        arg = torch.where(valid, arg, torch.full_like(arg, -1))
        ctx.save_for_backward(ref, qry, arg, wgt)
        ctx.n_ids, ctx.k = n_ids, k
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        ref, qry, arg, wgt = ctx.saved_tensors
        M0, C = ref.shape
        N = qry.shape[0]
        dev = qry.device
        need_ref, need_qry = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_ref or need_qry):
            return None, None, None, None, None
        g = grad_out.contiguous().float()
        gq_sum = gr_sum = None
        for j in range(ctx.k):
            gj = (g * wgt[j]).contiguous()
            gq = torch.empty((N, C), dtype=torch.float32, device=dev) if need_qry else None
            gr = torch.empty((M0, C), dtype=torch.float32, device=dev) if need_ref else None
            with torch.cuda.device(dev):
                rc = lib.manet_global_match_backward_f32(qry.data_ptr(), qry.stride(0), qry.stride(1), ref.data_ptr(),
                                                         ref.stride(0) if M0 > 0 else C, ref.stride(1) if M0 > 0 else 1,
                                                         arg[j].data_ptr(), gj.data_ptr(), N, M0, C, ctx.n_ids,
                                                         None if gq is None else gq.data_ptr(), C, 1,
                                                         None if gr is None else gr.data_ptr(), C, 1, _stream_ptr(dev))
            _lib.check(rc, "manet_global_match_backward_f32")
            if need_qry:
                gq_sum = gq if gq_sum is None else gq_sum.add_(gq)
            if need_ref:
                gr_sum = gr if gr_sum is None else gr_sum.add_(gr)
        return gr_sum, gq_sum, None, None, None


def _dense(t):
    """non-overlapping and dense (a permuted contiguous tensor): empty_strided can mirror its layout"""
    if t.numel() == 0:
        return False
    sizes_strides = sorted(zip(t.stride(), t.shape))
    expect = 1
    for st, sz in sizes_strides:
        if sz == 1:
            continue
        if st != expect:
            return False
        expect *= sz
    return True


class LocalMatchFn(torch.autograd.Function):
    """out [h, w, n_ids]; differentiable w.r.t. prev and cur [h, w, C] (any strides); downsample configuration."""

    @staticmethod
    def forward(ctx, prev, cur, labels, n_ids, max_distance):
        lib = _lib.load()
        h, w, C = cur.shape
        dev = cur.device
        P = 2 * max_distance + 1
        nbytes = ctypes.c_size_t(0)
        _lib.check(lib.manet_local_match_arg_workspace_bytes(h, w, C, max_distance, ctypes.byref(nbytes)),
                   "manet_local_match_arg_workspace_bytes")
        ws = _scratch(dev, nbytes.value)
        out = torch.empty((h, w, n_ids), dtype=torch.float32, device=dev)
        arg = torch.empty((h, w, n_ids), dtype=torch.int32, device=dev)
        vol = torch.empty((P * P, h // 2, w // 2), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.manet_local_match_arg_f32(prev.data_ptr(), prev.stride(0), prev.stride(1), prev.stride(2),
                                               cur.data_ptr(), cur.stride(0), cur.stride(1), cur.stride(2),
                                               labels.data_ptr(), h, w, C, n_ids, max_distance, out.data_ptr(),
                                               arg.data_ptr(), vol.data_ptr(), ws.data_ptr(), ws.numel(),
                                               _stream_ptr(dev))
        _lib.check(rc, "manet_local_match_arg_f32")
        ctx.save_for_backward(prev, cur, vol, arg)
        ctx.n_ids, ctx.max_distance = n_ids, max_distance
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        prev, cur, vol, arg = ctx.saved_tensors
        h, w, C = cur.shape
        dev = cur.device
        g = grad_out.contiguous().float()
        nbytes = ctypes.c_size_t(0)
        _lib.check(lib.manet_local_match_backward_workspace_bytes(h, w, C, ctx.max_distance, ctypes.byref(nbytes)),
                   "manet_local_match_backward_workspace_bytes")
        ws = _scratch(dev, nbytes.value)
        gp = torch.empty_strided(prev.shape, prev.stride(), dtype=torch.float32, device=dev) \
            if _dense(prev) else torch.empty(prev.shape, dtype=torch.float32, device=dev)
        gc = torch.empty_strided(cur.shape, cur.stride(), dtype=torch.float32, device=dev) \
            if _dense(cur) else torch.empty(cur.shape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.manet_local_match_backward_f32(prev.data_ptr(), prev.stride(0), prev.stride(1), prev.stride(2),
                                                    cur.data_ptr(), cur.stride(0), cur.stride(1), cur.stride(2),
                                                    vol.data_ptr(), arg.data_ptr(), g.data_ptr(), h, w, C, ctx.n_ids,
                                                    ctx.max_distance, gp.data_ptr(), gp.stride(0), gp.stride(1),
                                                    gp.stride(2), gc.data_ptr(), gc.stride(0), gc.stride(1),
                                                    gc.stride(2), ws.data_ptr(), ws.numel(), _stream_ptr(dev))
        _lib.check(rc, "manet_local_match_backward_f32")
        return gp, gc, None, None, None


class LocalMatchFullFn(torch.autograd.Function):
    """out [h, w, n_ids] for MODEL_LOCAL_DOWNSAMPLE = False (reference IntVOS.py:299-313 + :398-432): raw full-resolution
    window distances, masked minimum against the constant 1.0; the gradient of the min flows to one window offset per (pixel,
    object) -- none where the constant wins -- and from there to x - y of the two embeddings."""

    @staticmethod
    def forward(ctx, prev, cur, labels, n_ids, max_distance):
        lib = _lib.load()
        h, w, C = cur.shape
        dev = cur.device
        P = 2 * max_distance + 1
        out = torch.empty((h, w, n_ids), dtype=torch.float32, device=dev)
        arg = torch.empty((h, w, n_ids), dtype=torch.int32, device=dev)
        vol = torch.empty((h, w, P * P), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.manet_local_match_full_arg_f32(prev.data_ptr(), prev.stride(0), prev.stride(1), prev.stride(2),
                                                    cur.data_ptr(), cur.stride(0), cur.stride(1), cur.stride(2),
                                                    labels.data_ptr(), h, w, C, n_ids, max_distance, out.data_ptr(),
                                                    arg.data_ptr(), vol.data_ptr(), _stream_ptr(dev))
        _lib.check(rc, "manet_local_match_full_arg_f32")
        ctx.save_for_backward(prev, cur, arg)
        ctx.n_ids, ctx.max_distance = n_ids, max_distance
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        prev, cur, arg = ctx.saved_tensors
        h, w, C = cur.shape
        dev = cur.device
        P = 2 * ctx.max_distance + 1
        g = grad_out.contiguous().float()
        pc, cc = prev.permute(2, 0, 1).contiguous(), cur.permute(2, 0, 1).contiguous()  # (no copy for C-major embeddings)
        gp, gc = torch.empty_like(pc), torch.empty_like(cc)
        dv = torch.empty((P * P, h * w), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.manet_local_match_full_backward_f32(pc.data_ptr(), cc.data_ptr(), arg.data_ptr(), g.data_ptr(), h, w, C,
                                                         ctx.n_ids, ctx.max_distance, gp.data_ptr(), gc.data_ptr(),
                                                         dv.data_ptr(), _stream_ptr(dev))
        _lib.check(rc, "manet_local_match_full_backward_f32")
        return gp.permute(1, 2, 0), gc.permute(1, 2, 0), None, None, None


class CorrelationFn(torch.autograd.Function):
    """correlation_package's CorrelationFunction (correlation.py:7-45) on the HIP kernels.  The output (and the gradients)
    have the inputs' dtype, as the reference's op (float / half / double: its dispatch, correlation_cuda_kernel.cu:495-541; the
    forward computes in that type exactly like the no-grad path).  The backward kernels are fp32 and fp64: half gradients are
    computed in fp32 and rounded once, double gradients in double."""

    @staticmethod
    def forward(ctx, input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2):
        from . import ops
        if input1.dtype != input2.dtype:
            raise RuntimeError("cvpr2020_manet_amd: correlation inputs must have the same dtype")
        a = input1.contiguous()
        b = input2.contiguous()
        ctx.save_for_backward(a, b)
        ctx.params = (pad_size, kernel_size, max_displacement, stride1, stride2)
        with torch.no_grad():
            return ops.correlation_forward(a, b, pad_size, kernel_size, max_displacement, stride1, stride2)

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        a0, b0 = ctx.saved_tensors
        wide = torch.float64 if a0.dtype == torch.float64 else torch.float32
        a, b = a0.to(wide), b0.to(wide)
        B, C, H, W = a.shape
        pad_size, kernel_size, max_displacement, stride1, stride2 = ctx.params
        g = grad_out.contiguous().to(wide)
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        fn = lib.manet_correlation_backward_f64 if wide == torch.float64 else lib.manet_correlation_backward_f32
        with torch.cuda.device(a.device):
            rc = fn(a.data_ptr(), b.data_ptr(), g.data_ptr(), B, C, H, W, pad_size, kernel_size, max_displacement, stride1,
                    stride2, ga.data_ptr(), gb.data_ptr(), _stream_ptr(a.device))
        _lib.check(rc, "manet_correlation_backward")
        return ga.to(a0.dtype), gb.to(b0.dtype), None, None, None, None, None
