"""torch-tensor front-end of the C ABI (include/manet_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every function
below hands raw device pointers, element strides and that stream to libmanet_hip.so.  All inputs
must live on a HIP device -- there is deliberately no CPU path (a silent fallback would void the
parity claims); CPU tensors raise.
"""
import torch

from . import _lib

# "bf16r": bf16 filter + exact fp32 re-rank (MANET_COMPUTE_BF16_REFINE): MANET_COMPUTE_F32's result bit for bit, ~1.4x bf16's cost
COMPUTE = {"f32": _lib.COMPUTE_F32, "fp32": _lib.COMPUTE_F32, "bf16": _lib.COMPUTE_BF16,
           "bf16x3": _lib.COMPUTE_BF16X3, "bf16r": _lib.COMPUTE_BF16_REFINE}


def _image_kind(compute_code):
    """operand images depend on the arithmetic only through this: bf16r packs exactly as bf16"""
    return _lib.COMPUTE_BF16 if compute_code == _lib.COMPUTE_BF16_REFINE else compute_code

_ws_cache = {}


def _need_gpu(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("cvpr2020_manet_amd: %s must be a tensor on a HIP device "
                           "(the matching path has no CPU fallback)" % name)


def _refuse_autograd(what, *tensors):
    """The HIP ops below hand raw pointers to the C ABI and return tensors without a grad_fn.  The
    reference's functions are differentiable (train_stage1.py:126-156 back-propagates through them), so
    silently detaching would train the heads with zero gradient through the match maps: raise instead.
    Ops that do have a backward (`global_match`, `local_match`, `correlation_forward`) route through their
    torch.autograd.Function in cvpr2020_manet_amd.autograd instead of coming here."""
    if not torch.is_grad_enabled():
        return
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.requires_grad:
            raise RuntimeError("cvpr2020_manet_amd.ops.%s: an input requires grad, but this op has no "
                               "backward -- call it under torch.no_grad() or detach the input "
                               "(gradients would otherwise be dropped silently)" % what)


def _wants_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def _global_match_autograd(reference_embeddings, query_embeddings, reference_labels, n_ids, k_nearest_neighbors,
                           compute, normalize, mem):
    """Training route of global_match (train_stage1.py:126-156 back-propagates through it): the arg-min forward +
    explicit backward of autograd.GlobalMatchFn; normalisation (IntVOS.py:611-612) and the min-merge with the
    stored map (:620-622, the stored copy detached as in the reference) are ordinary differentiable torch ops."""
    from .autograd import GlobalMatchFn, GlobalMatchTopkFn
    if COMPUTE[compute] != _lib.COMPUTE_F32 or not 1 <= k_nearest_neighbors <= 8:
        raise RuntimeError("cvpr2020_manet_amd.ops.global_match: the backward exists for compute='f32' and "
                           "k_nearest_neighbors 1..8 only (got k=%d, compute=%r)" % (k_nearest_neighbors, compute))
    ref, M0, C = _flat(reference_embeddings, "reference_embeddings")
    qry, N, C2 = _flat(query_embeddings, "query_embeddings")
    if C != C2:
        raise ValueError("embedding_dim mismatch: %d vs %d" % (C, C2))
    lab = _labels(reference_labels, "reference_labels")
    if lab.numel() != M0:
        raise ValueError("reference_labels has %d entries for %d reference pixels" % (lab.numel(), M0))
    ref, qry = ref.float(), qry.float()  # (a differentiable widening if the embeddings are stored in bf16)
    if k_nearest_neighbors > 1:
        if M0 < k_nearest_neighbors:
            raise RuntimeError("selected index k out of range")  # what torch.topk raises (IntVOS.py:87)
        out = GlobalMatchTopkFn.apply(ref, qry, lab, n_ids, int(k_nearest_neighbors))
    else:
        out, _ = GlobalMatchFn.apply(ref, qry, lab, n_ids)
    if normalize:
        out = (torch.sigmoid(out) - 0.5) * 2
    if mem is not None:
        m = mem.view_as(out)
        out = torch.where(out <= m, out, m)
        with torch.no_grad():
            m.copy_(out)
    return out


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


class _NoSwitch:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(device):
    """`with _on(dev):` = `with _on(dev):`, minus the guard object when `dev` already is the current device (the
    usual case: ~4 us of host time per call, 20 calls per propagated frame -- the eager loop is host-bound)"""
    return _NO_SWITCH if torch.cuda.current_device() == device.index else torch.cuda.device(device)


def _workspace(device, tag, nbytes):
    """Per (device, stream, tag) scratch tensor, grown on demand (torch allocator = plumbing)."""
    key = (device.index, _stream_ptr(device), tag)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def _armed_workspace(device, tag, nbytes):
    """Like _workspace, but the tensor is created FILLED with 0xff bytes and is only ever handed to calls that keep it so
    (MANET_EPI_KEYS_ARMED: the match keys are re-armed by the epilogue that reads them) -- no per-call fill launch."""
    key = (device.index, _stream_ptr(device), tag)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.full((int(nbytes * 1.25) + 1024,), 0xff, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def _emb(t):
    """embeddings are consumed in the producer's storage: float32, or bfloat16 (2-byte reads end to end,
    SURVEY.md 8f rank 4); anything else is widened to float32"""
    return t if t.dtype in (torch.float32, torch.bfloat16) else t.float()


def _emb_code(t):
    return _lib.EMB_BF16 if t.dtype == torch.bfloat16 else _lib.EMB_F32


def _flat(t, name):
    """[..., C] -> ([rows, C] view, rows, C); like the reference's .view(-1, C) (IntVOS.py:203-204)
    but a non-viewable input is copied instead of raising."""
    _need_gpu(t, name)
    t = _emb(t)
    t2 = t.reshape(-1, t.shape[-1])
    return t2, t2.shape[0], t2.shape[1]


def _labels(t, name):
    _need_gpu(t, name)
    return t.reshape(-1).to(torch.int32).contiguous()


def global_match(reference_embeddings, query_embeddings, reference_labels, n_ids, k_nearest_neighbors=1,
                 compute="f32", normalize=False, mem=None):
    """nearest_neighbor_features_per_object on the HIP path (IntVOS.py:160-210), optionally with the
    fused normalise (IntVOS.py:611-612) and min-merge into `mem` (IntVOS.py:620-622, in place).

    Returns float32 [N, n_ids] (N = number of query pixels), raw or normalised distances.
    """
    import ctypes
    if _wants_grad(reference_embeddings, query_embeddings, mem):
        return _global_match_autograd(reference_embeddings, query_embeddings, reference_labels, n_ids,
                                      k_nearest_neighbors, compute, normalize, mem)
    lib = _lib.load()
    ref, M0, C = _flat(reference_embeddings, "reference_embeddings")
    qry, N, C2 = _flat(query_embeddings, "query_embeddings")
    if C != C2:
        raise ValueError("embedding_dim mismatch: %d vs %d" % (C, C2))
    lab = _labels(reference_labels, "reference_labels")
    if lab.numel() != M0:
        raise ValueError("reference_labels has %d entries for %d reference pixels" % (lab.numel(), M0))
    if k_nearest_neighbors > 1 and M0 < k_nearest_neighbors:
        raise RuntimeError("selected index k out of range")  # what torch.topk raises (IntVOS.py:87)
    dev = qry.device
    cmp_ = COMPUTE[compute]
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.manet_global_match_workspace_bytes(N, M0, C, n_ids, k_nearest_neighbors, cmp_,
                                                      ctypes.byref(nbytes)), "manet_global_match_workspace_bytes")
    ws = _workspace(dev, "global", nbytes.value)
    out = torch.empty((N, n_ids), dtype=torch.float32, device=dev)
    mem_ptr = None
    if mem is not None:
        _need_gpu(mem, "mem")
        if mem.dtype != torch.float32 or not mem.is_contiguous() or mem.numel() != N * n_ids:
            raise ValueError("mem must be a contiguous float32 tensor of N*n_ids elements")
        mem_ptr = mem.data_ptr()
    flags = _lib.EPI_NORMALIZE if normalize else 0
    with _on(dev):
        rc = lib.manet_global_match_ex(qry.data_ptr(), _emb_code(qry), qry.stride(0), qry.stride(1), ref.data_ptr(),
                                       _emb_code(ref), ref.stride(0) if M0 > 0 else C, ref.stride(1) if M0 > 0 else 1,
                                       lab.data_ptr(), N, M0, C, n_ids, k_nearest_neighbors, cmp_,
                                       out.data_ptr(), mem_ptr, flags, ws.data_ptr(), ws.numel(),
                                       _stream_ptr(dev))
    _lib.check(rc, "manet_global_match_ex")
    return out


_PROBE_POOL = {"block": None, "next": 0}


def _probe_slot():
    """Two pinned int32 the adaptive probe kernel of compute="bf16r" writes through a raw pointer.  Slots of ONE pinned block
    that lives as long as the process (a per-bank pin_memory() tensor could go back to torch's pinned allocator
    with a device write still in flight -- the allocator knows nothing of it); handed out round robin, 1024 of them."""
    if _PROBE_POOL["block"] is None:
        _PROBE_POOL["block"] = torch.zeros(1024, 2, dtype=torch.int32).pin_memory()
    i = _PROBE_POOL["next"]
    _PROBE_POOL["next"] = (i + 1) % 1024
    return _PROBE_POOL["block"][i]


class PreparedBank:
    """A memory bank sorted by object id and packed for the MFMA loop, reusable across frames
    (test.py:237-259 matches every frame of a clip against the same annotated frame)."""

    def __init__(self, reference_embeddings, reference_labels, n_ids, compute="f32", reuse=None):
        """reuse: a PreparedBank that is no longer needed -- its workspace is taken over when it is large enough (a new
        interaction round re-prepares a bank of the same size: no 50-100 MB allocation per round)"""
        import ctypes
        _refuse_autograd("PreparedBank", reference_embeddings)
        lib = _lib.load()
        ref, M0, C = _flat(reference_embeddings, "reference_embeddings")
        lab = _labels(reference_labels, "reference_labels")
        if lab.numel() != M0:
            raise ValueError("reference_labels has %d entries for %d reference pixels" % (lab.numel(), M0))
        self.M0, self.C, self.n_ids, self.compute = M0, C, n_ids, COMPUTE[compute]
        self.device = ref.device
        nbytes = ctypes.c_size_t(0)
        _lib.check(lib.manet_bank_workspace_bytes(M0, C, n_ids, self.compute, ctypes.byref(nbytes)),
                   "manet_bank_workspace_bytes")
        if getattr(reuse, "_adapt", None) is not None:
            self._adapt = reuse._adapt  # (what the previous bank of this clip learnt about its embeddings carries over)
        old = getattr(reuse, "ws", None)
        if old is not None and old.device == self.device and old.numel() >= nbytes.value:
            self.ws, reuse.ws = old, None
        else:
            self.ws = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
        with _on(self.device):
            rc = lib.manet_bank_prepare_ex(ref.data_ptr(), _emb_code(ref), ref.stride(0) if M0 > 0 else C,
                                           ref.stride(1) if M0 > 0 else 1, lab.data_ptr(), M0, C, n_ids,
                                           self.compute, self.ws.data_ptr(), self.ws.numel(),
                                           _stream_ptr(self.device))
        _lib.check(rc, "manet_bank_prepare_ex")

    # compute="bf16r", adaptive policy: a frame whose filter pass sent more than this share of its query tiles to the rescue
    # pass (the exact fp32 kernel) cost the filter AND most of the fp32 kernel; the next ADAPT_FRAMES frames then skip the
    # filter (MANET_EPI_REFINE_EXACT: the same bits at the fp32 path's cost), after which one frame probes the filter again.
    # (The rescue launch is dealt to the rescued tiles only: a share r costs ~r of the fp32 kernel, so filtering pays below
    # r ~ 1 - filter / fp32 kernel ~ 0.85 at cfg2 shape; r3's all-or-nothing rescue had the switch at one half.)
    ADAPT_SHARE, ADAPT_FRAMES, ADAPT_PROBE_EVERY = 0.8, 16, 4

    def match(self, query_embeddings, k_nearest_neighbors=1, normalize=False, mem=None, out=None, adaptive=True):
        """query_embeddings: [..., C] float32 / bfloat16 tensor, or a PackedQuery (operand image made once,
        e.g. for every frame of a clip right after extract_feature: no per-frame pack pass).
        adaptive (compute="bf16r" only): let the previous frames' rescue share decide whether this frame runs the bf16 filter at
        all (see ADAPT_SHARE); the result is the fp32 kernel's bit for bit either way.  The share arrives by an asynchronous copy
        and is read a frame late -- no synchronisation; nothing adapts inside a HIP-graph capture."""
        import ctypes
        lib = _lib.load()
        armed, raw = False, None
        if isinstance(query_embeddings, (PackedQuery, PreparedFrame)):
            pq = query_embeddings
            if pq.C != self.C or _image_kind(pq.compute) != _image_kind(self.compute):
                raise ValueError("the query operand was packed for C=%d compute=%d, bank has C=%d compute=%d"
                                 % (pq.C, pq.compute, self.C, self.compute))
            _refuse_autograd("PreparedBank.match", mem)
            qry, N, C, q_code, q_s0, q_s1 = pq.image, pq.N, pq.C, _lib.EMB_PACKED, 0, 0
            armed = k_nearest_neighbors == 1 and self.compute != _lib.COMPUTE_BF16_REFINE
            if self.compute == _lib.COMPUTE_BF16_REFINE:  # the exact re-rank reads the query as stored
                if pq.raw is None:
                    raise ValueError("compute='bf16r' needs the embedding the operand image was made from (still alive)")
                raw, _, _ = _flat(pq.raw, "query_embeddings")
        else:
            _refuse_autograd("PreparedBank.match", query_embeddings, mem)
            qry, N, C = _flat(query_embeddings, "query_embeddings")
            q_code, q_s0, q_s1 = _emb_code(qry), qry.stride(0), qry.stride(1)
        if C != self.C:
            raise ValueError("embedding_dim mismatch: %d vs %d" % (C, self.C))
        if k_nearest_neighbors > 1 and self.M0 < k_nearest_neighbors:
            raise RuntimeError("selected index k out of range")
        dev = qry.device if isinstance(qry, torch.Tensor) else self.device
        nbytes = ctypes.c_size_t(0)
        _lib.check(lib.manet_match_workspace_bytes(N, self.M0, C, self.n_ids, k_nearest_neighbors,
                                                   self.compute, ctypes.byref(nbytes)),
                   "manet_match_workspace_bytes")
        ws = _armed_workspace(dev, "match_armed", nbytes.value) if armed else _workspace(dev, "match", nbytes.value)
        if out is None:
            out = torch.empty((N, self.n_ids), dtype=torch.float32, device=dev)
        mem_ptr = None
        if mem is not None:
            _need_gpu(mem, "mem")
            if mem.dtype != torch.float32 or not mem.is_contiguous() or mem.numel() != N * self.n_ids:
                raise ValueError("mem must be a contiguous float32 tensor of N*n_ids elements")
            mem_ptr = mem.data_ptr()
        flags = (_lib.EPI_NORMALIZE if normalize else 0) | (_lib.EPI_KEYS_ARMED if armed else 0)
        self._last = (ws, N)  # (refine_stats)
        refine = self.compute == _lib.COMPUTE_BF16_REFINE
        adapt = refine and adaptive and not torch.cuda.is_current_stream_capturing()
        forced = False
        if adapt:
            ad = self.__dict__.setdefault("_adapt", {"host": None, "event": None, "left": 0, "seen": 0})
            if ad["event"] is not None and ad["event"].query():  # the share of a frame or two ago has arrived
                rescued, tiles = int(ad["host"][0]), int(ad["host"][1])
                ad["event"] = None
                if tiles > 0 and rescued > self.ADAPT_SHARE * tiles:
                    ad["left"] = self.ADAPT_FRAMES
            forced = ad["left"] > 0
            if forced:
                ad["left"] -= 1
                flags |= _lib.EPI_REFINE_EXACT
        self.last_match_forced_exact = forced
        with _on(dev):
            if raw is not None:
                rc = lib.manet_global_match_refine(raw.data_ptr(), _emb_code(raw), raw.stride(0), raw.stride(1), qry.data_ptr(),
                                                   self.ws.data_ptr(), N, self.M0, C, self.n_ids, out.data_ptr(), mem_ptr,
                                                   flags, ws.data_ptr(), ws.numel(), _stream_ptr(dev))
            else:
                rc = lib.manet_global_match_prepared_ex(qry.data_ptr(), q_code, q_s0, q_s1,
                                                        self.ws.data_ptr(), N, self.M0, C, self.n_ids,
                                                        k_nearest_neighbors, self.compute, out.data_ptr(), mem_ptr,
                                                        flags, ws.data_ptr(), ws.numel(), _stream_ptr(dev))
        if adapt and not forced:
            ad["seen"] = ad.get("seen", 0) + 1
        if rc == 0 and adapt and not forced and ad["event"] is None and (ad["seen"] - 1) % self.ADAPT_PROBE_EVERY == 0:
            # this frame's rescue share: one tiny launch counts it straight into pinned host memory (device-visible), an event
            # says when it is there -- looked at by a later call, never waited for.  Every ADAPT_PROBE_EVERY-th filtered frame
            # only: a probe costs ~10 us of stream time, and what the share says changes with the clip, not with the frame.
            if ad["host"] is None:
                ad["host"] = _probe_slot()
            with _on(dev):
                rc2 = lib.manet_global_match_refine_rescued_async(ws.data_ptr(), N, C, self.n_ids, ad["host"].data_ptr(),
                                                                  _stream_ptr(dev))
            _lib.check(rc2, "manet_global_match_refine_rescued_async")
            ad["event"] = torch.cuda.Event()
            ad["event"].record(torch.cuda.current_stream(dev))  # (the stream the launches above went to: _stream_ptr(dev))
        if rc != 0 and armed:
            # the armed workspace is only all-0xff again once the finish kernel has run: after a failed call nothing is
            # known about it -- drop it, the next armed call fills a fresh one
            _ws_cache.pop((dev.index, _stream_ptr(dev), "match_armed"), None)
        _lib.check(rc, "manet_global_match_prepared_ex")
        return out

    def refine_stats(self):
        """compute='bf16r': (candidate rows the filter pass appended, 1 if the candidate list overflowed and every pair
        scanned its object's rows instead) of the LAST match() on this bank; synchronises."""
        import ctypes
        if self.compute != _lib.COMPUTE_BF16_REFINE or getattr(self, "_last", None) is None:
            raise RuntimeError("refine_stats: no compute='bf16r' match has run on this bank")
        ws, N = self._last
        torch.cuda.current_stream(ws.device).synchronize()  # (the C call copies on the null stream, which torch's side streams do not block)
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(_lib.load().manet_global_match_refine_stats(ws.data_ptr(), N, self.C, self.n_ids, ctypes.byref(a),
                                                               ctypes.byref(b)), "manet_global_match_refine_stats")
        return a.value, b.value

    def refine_stats_full(self):
        """compute='bf16r', the LAST match() on this bank (synchronises): dict with the candidate rows the filter pass
        listed, the overflow flag, and how many of the frame's 256-query tiles went through the rescue pass (the exact fp32
        kernel) -- `rescued_tile_fraction` is the distribution-dependent part of this mode's cost.  (The figures live in the
        stream-shared match workspace: read them before another bf16r match runs on this stream.)"""
        import ctypes
        if self.compute != _lib.COMPUTE_BF16_REFINE or getattr(self, "_last", None) is None:
            raise RuntimeError("refine_stats: no compute='bf16r' match has run on this bank")
        ws, N = self._last
        torch.cuda.current_stream(ws.device).synchronize()  # (as refine_stats)
        s4 = (ctypes.c_int64 * 4)()
        _lib.check(_lib.load().manet_global_match_refine_stats2(ws.data_ptr(), N, self.C, self.n_ids, s4),
                   "manet_global_match_refine_stats2")
        return {"candidate_rows": int(s4[0]), "candidate_rows_per_pair": s4[0] / float(N * self.n_ids),
                "list_overflowed": int(s4[1]), "rescued_tiles": int(s4[2]), "query_tiles": int(s4[3]),
                "rescued_tile_fraction": s4[2] / float(max(1, s4[3]))}


class PackedQuery:
    """The query operand image of ONE frame (manet_query_pack): made once per frame of a clip -- the reference
    computes all embeddings of a clip up front (test.py:143-154) -- and matched against any number of banks /
    interaction rounds without a per-frame pack pass."""

    def __init__(self, query_embeddings, compute="f32"):
        import ctypes
        _refuse_autograd("PackedQuery", query_embeddings)
        lib = _lib.load()
        qry, N, C = _flat(query_embeddings, "query_embeddings")
        self.N, self.C, self.compute = N, C, COMPUTE[compute]
        nbytes = ctypes.c_size_t(0)
        _lib.check(lib.manet_query_pack_bytes(N, C, self.compute, ctypes.byref(nbytes)), "manet_query_pack_bytes")
        self.image = torch.empty(nbytes.value, dtype=torch.uint8, device=qry.device)
        with _on(qry.device):
            rc = lib.manet_query_pack(qry.data_ptr(), _emb_code(qry), qry.stride(0), qry.stride(1), N, C, self.compute,
                                      self.image.data_ptr(), self.image.numel(), _stream_ptr(qry.device))
        _lib.check(rc, "manet_query_pack")
        self.device = qry.device
        self.raw = query_embeddings  # (compute="bf16r": the exact re-rank reads the embedding itself)


class PreparedFrame:
    """The per-frame operands of ONE frame (manet_frame_prepare): the query operand image of the global match and, when
    `max_distance` >= 0, the padded 2x2-pooled plane (+ tile table) of the local match -- made from one read of the
    embedding, once per frame: a propagated frame then needs no pack pass and no pooling pass, and the previous frame's
    embedding is not read at all (its plane was made when it was the current frame, test.py:259)."""

    def __init__(self, ws, h, w, C, compute, max_distance, keep=None, raw=None):
        self.ws, self.h, self.w, self.C, self.compute, self.max_distance = ws, h, w, C, compute, max_distance
        self.raw = raw  # the [C, h, w] embedding as an [h, w, C] view (compute="bf16r" re-ranks from it)
        self.N = h * w
        self.image = ws  # the operand image sits at the start of the frame workspace (a MANET_EMB_PACKED query)
        self.device = ws.device
        self.keep = keep  # whatever must stay alive with this object (the identity-keyed caches key on storage pointers)


def frame_workspace_bytes(h, w, C, compute="f32", max_distance=-1):
    import ctypes
    nbytes = ctypes.c_size_t(0)
    _lib.check(_lib.load().manet_frame_workspace_bytes(h, w, C, COMPUTE[compute], max_distance, ctypes.byref(nbytes)),
               "manet_frame_workspace_bytes")
    return nbytes.value


def prepare_frames(embeddings, compute="f32", max_distance=-1, preset=None, preset_value=1.0):
    """manet_frame_prepare on `embeddings` = [C, h, w] (one frame) or [B, C, h, w] (extract_feature's batch,
    IntVOS.py:578-581) in float32 / bfloat16 storage, any strides: ONE launch for the whole batch.  Returns a
    PreparedFrame (one frame) or a list of them.  `preset`: an optional contiguous float32 tensor set to `preset_value`
    by the same launch (the next local match's `out`, see local_match_frames)."""
    lib = _lib.load()
    _need_gpu(embeddings, "embeddings")
    _refuse_autograd("prepare_frames", embeddings)
    emb = _emb(embeddings)
    single = emb.dim() == 3
    if single:
        emb = emb.unsqueeze(0)
    if emb.dim() != 4:
        raise ValueError("embeddings must be [C, h, w] or [B, C, h, w]")
    B, C, h, w = emb.shape
    if B == 0:
        return []
    cmp_ = COMPUTE[compute]
    per = frame_workspace_bytes(h, w, C, compute, max_distance)
    ws = torch.empty((B, per), dtype=torch.uint8, device=emb.device)
    fill_ptr, fill_words, fill_bits = None, 0, 0
    if preset is not None:
        _need_gpu(preset, "preset")
        if preset.dtype != torch.float32 or not preset.is_contiguous():
            raise ValueError("preset must be a contiguous float32 tensor")
        import struct
        fill_ptr, fill_words = preset.data_ptr(), preset.numel()
        fill_bits = struct.unpack("<I", struct.pack("<f", float(preset_value)))[0]
    with _on(emb.device):
        rc = lib.manet_frame_prepare(emb.data_ptr(), _emb_code(emb), emb.stride(0), emb.stride(2), emb.stride(3),
                                     emb.stride(1), B, h, w, C, cmp_, max_distance, ws.data_ptr(), per, fill_ptr,
                                     fill_words, fill_bits, _stream_ptr(emb.device))
    _lib.check(rc, "manet_frame_prepare")
    frames = [PreparedFrame(ws[i], h, w, C, cmp_, max_distance, raw=emb[i].permute(1, 2, 0)) for i in range(B)]
    return frames[0] if single else frames


def embed_finish(conv_out, scale, shift, relu=True, emb_dtype=torch.float32, compute="f32", max_distance=-1):
    """manet_embed_finish: the embedding layer's epilogue fused with the frame prepare (IntVOS.py:537-543, :578-581).
    conv_out [B, C, h, w] float32 = embedding_conv's raw output; scale / shift [C] = eval-mode bn2 folded (ops.fold_bn).
    ONE launch -> (embedding [B, C, h, w] in emb_dtype = relu(conv_out * scale + shift), list of PreparedFrames made from the
    embedding as stored -- bit-identical to prepare_frames(embedding))."""
    lib = _lib.load()
    _need_gpu(conv_out, "conv_out")
    _refuse_autograd("embed_finish", conv_out, scale, shift)
    if conv_out.dim() != 4 or conv_out.dtype != torch.float32:
        raise ValueError("conv_out must be float32 [B, C, h, w]")
    B, C, h, w = conv_out.shape
    if emb_dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("emb_dtype must be torch.float32 or torch.bfloat16")
    scale, shift = scale.detach().float().contiguous(), shift.detach().float().contiguous()
    if scale.numel() != C or shift.numel() != C:
        raise ValueError("scale / shift must have C elements")
    cmp_ = COMPUTE[compute]
    emb = torch.empty((B, C, h, w), dtype=emb_dtype, device=conv_out.device)
    if B == 0:
        return emb, []
    per = frame_workspace_bytes(h, w, C, compute, max_distance)
    ws = torch.empty((B, per), dtype=torch.uint8, device=conv_out.device)
    with _on(conv_out.device):
        rc = lib.manet_embed_finish(conv_out.data_ptr(), conv_out.stride(0), conv_out.stride(2), conv_out.stride(3),
                                    conv_out.stride(1), scale.data_ptr(), shift.data_ptr(), int(bool(relu)), emb.data_ptr(),
                                    _emb_code(emb), B, h, w, C, cmp_, max_distance, ws.data_ptr(), per, _stream_ptr(conv_out.device))
    _lib.check(rc, "manet_embed_finish")
    frames = [PreparedFrame(ws[i], h, w, C, cmp_, max_distance, raw=emb[i].permute(1, 2, 0)) for i in range(B)]
    return emb, frames


def local_match_frames(prev_frame, cur_frame, prev_frame_labels, n_ids, out=None, out_is_preset=False):
    """local_previous_frame_nearest_neighbor_features_per_object (IntVOS.py:345-434, downsample configuration) on two
    PreparedFrames: the fused window / min kernel alone -> [h, w, n_ids]; bit-identical to local_match on the
    embeddings the frames were prepared from."""
    lib = _lib.load()
    a, b = prev_frame, cur_frame
    if (a.h, a.w, a.C, _image_kind(a.compute), a.max_distance) != (b.h, b.w, b.C, _image_kind(b.compute), b.max_distance):
        raise ValueError("the two frames were prepared for different shapes / arithmetic / window radius")
    if b.max_distance < 0:
        raise ValueError("the frames were prepared without a pooled plane (max_distance < 0)")
    lab = _labels(prev_frame_labels, "prev_frame_labels")
    if lab.numel() != b.h * b.w:
        raise ValueError("prev_frame_labels must have height*width entries")
    dev = b.device
    if out is None:
        out = torch.empty((b.h, b.w, n_ids), dtype=torch.float32, device=dev)
        out_is_preset = False
    elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != b.h * b.w * n_ids:
        raise ValueError("out must be a contiguous float32 tensor of h*w*n_ids elements")
    with _on(dev):
        rc = lib.manet_local_match_frames(a.ws.data_ptr(), b.ws.data_ptr(), lab.data_ptr(), b.h, b.w, b.C, b.compute,
                                          n_ids, b.max_distance, out.data_ptr(), int(bool(out_is_preset)),
                                          _stream_ptr(dev))
    _lib.check(rc, "manet_local_match_frames")
    return out.view(b.h, b.w, n_ids)


def local_volume_bytes(h, w, max_distance):
    """bytes of one frame pair's stored window-distance volume (manet_local_volume_bytes)"""
    import ctypes
    nbytes = ctypes.c_size_t(0)
    _lib.check(_lib.load().manet_local_volume_bytes(h, w, max_distance, ctypes.byref(nbytes)), "manet_local_volume_bytes")
    return nbytes.value


def local_volumes(prev_frames, cur_frames, out=None):
    """The label-independent half of the local match (IntVOS.py:266-296: pooled frames -> (2d+1)^2 window distances ->
    (sigmoid - 0.5) * 2) for a BATCH of frame pairs (prev_frames[i], cur_frames[i]) -- lists of PreparedFrames of one geometry --
    in ceil(n / 32) launches.  Returns a float32 tensor [n, floats per volume]; row i feeds local_match_volume.  `out`: an
    optional contiguous float32 tensor of that shape to write into."""
    import ctypes
    lib = _lib.load()
    n = len(cur_frames)
    if len(prev_frames) != n:
        raise ValueError("prev_frames and cur_frames must pair up")
    if n == 0:
        return torch.empty((0, 0), dtype=torch.float32)
    b = cur_frames[0]
    for a in list(prev_frames) + list(cur_frames):
        if (a.h, a.w, a.C, _image_kind(a.compute), a.max_distance, a.device) != (b.h, b.w, b.C, _image_kind(b.compute),
                                                                               b.max_distance, b.device):
            raise ValueError("the frames were prepared for different shapes / arithmetic / window radius / devices")
    if b.max_distance < 0:
        raise ValueError("the frames were prepared without a pooled plane (max_distance < 0)")
    per = local_volume_bytes(b.h, b.w, b.max_distance) // 4
    if out is None:
        out = torch.empty((n, per), dtype=torch.float32, device=b.device)
    elif out.dtype != torch.float32 or not out.is_contiguous() or tuple(out.shape) != (n, per) or out.device != b.device:
        raise ValueError("out must be a contiguous float32 [%d, %d] tensor on the frames' device" % (n, per))
    arr = ctypes.c_void_p * n
    pv = arr(*[f.ws.data_ptr() for f in prev_frames])
    cv = arr(*[f.ws.data_ptr() for f in cur_frames])
    vv = arr(*[out[i].data_ptr() for i in range(n)])
    with _on(b.device):
        rc = lib.manet_local_volume_frames(pv, cv, vv, n, b.h, b.w, b.C, b.compute, b.max_distance, _stream_ptr(b.device))
    _lib.check(rc, "manet_local_volume_frames")
    return out


def local_match_volume(volume, cur_frame, prev_frame_labels, n_ids, out=None, out_is_preset=False):
    """The label-dependent tail of the local match (IntVOS.py:398-432) on a stored volume (one row of local_volumes) -> [h, w,
    n_ids]; bit-identical to local_match_frames(prev_frame, cur_frame, ...) on the pair the volume was made from."""
    lib = _lib.load()
    b = cur_frame
    if b.max_distance < 0:
        raise ValueError("the frame was prepared without a pooled plane (max_distance < 0)")
    if (volume.dtype != torch.float32 or not volume.is_contiguous() or volume.device != b.device
            or volume.numel() * 4 != local_volume_bytes(b.h, b.w, b.max_distance)):
        raise ValueError("volume must be one contiguous float32 row of local_volumes() for this frame geometry")
    lab = _labels(prev_frame_labels, "prev_frame_labels")
    if lab.numel() != b.h * b.w:
        raise ValueError("prev_frame_labels must have height*width entries")
    dev = b.device
    if out is None:
        out = torch.empty((b.h, b.w, n_ids), dtype=torch.float32, device=dev)
        out_is_preset = False
    elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != b.h * b.w * n_ids:
        raise ValueError("out must be a contiguous float32 tensor of h*w*n_ids elements")
    with _on(dev):
        rc = lib.manet_local_match_volume(volume.data_ptr(), b.ws.data_ptr(), lab.data_ptr(), b.h, b.w, b.C, b.compute, n_ids,
                                          b.max_distance, out.data_ptr(), int(bool(out_is_preset)), _stream_ptr(dev))
    _lib.check(rc, "manet_local_match_volume")
    return out.view(b.h, b.w, n_ids)


def normalize_merge_(x, mem=None, normalize=True):
    """In place: x = (sigmoid(x)-0.5)*2 if normalize; if mem: x = mem = min(x, mem)
    (IntVOS.py:611-612, :620-622, :718-723)."""
    _refuse_autograd("normalize_merge_", x, mem)
    lib = _lib.load()
    _need_gpu(x, "x")
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("x must be contiguous float32")
    mem_ptr = None
    if mem is not None:
        _need_gpu(mem, "mem")
        if mem.dtype != torch.float32 or not mem.is_contiguous() or mem.numel() != x.numel():
            raise ValueError("mem must be contiguous float32 with x's element count")
        mem_ptr = mem.data_ptr()
    with _on(x.device):
        rc = lib.manet_normalize_merge_f32(x.data_ptr(), mem_ptr, x.numel(), int(bool(normalize)),
                                           _stream_ptr(x.device))
    _lib.check(rc, "manet_normalize_merge_f32")
    return x


def _hwc(t, name, allow_bf16=False):
    _need_gpu(t, name)
    if t.dim() != 3:
        raise ValueError("%s must be [height, width, embedding_dim]" % name)
    if allow_bf16 and t.dtype == torch.bfloat16:
        return t
    return t if t.dtype == torch.float32 else t.float()


def local_dist(x, y, max_distance, downsample=True):
    """local_pairwise_distances2(x=query, y=prev) (IntVOS.py:266-315) -> [h, w, (2d+1)^2]."""
    import ctypes
    lib = _lib.load()
    _refuse_autograd("local_dist", x, y)
    x = _hwc(x, "x")
    y = _hwc(y, "y")
    h, w, C = x.shape
    if tuple(y.shape) != (h, w, C):
        raise ValueError("x and y must have the same shape")
    dev = x.device
    P = 2 * max_distance + 1
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.manet_local_workspace_bytes(h, w, C, max_distance, int(bool(downsample)),
                                               ctypes.byref(nbytes)), "manet_local_workspace_bytes")
    ws = _workspace(dev, "local", nbytes.value)
    out = torch.empty((h, w, P * P), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = lib.manet_local_dist_f32(x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), y.data_ptr(),
                                      y.stride(0), y.stride(1), y.stride(2), h, w, C, max_distance,
                                      int(bool(downsample)), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                      _stream_ptr(dev))
    _lib.check(rc, "manet_local_dist_f32")
    return out


def local_match(prev_frame_embedding, query_embedding, prev_frame_labels, n_ids, max_distance=12,
                downsample=True):
    """local_previous_frame_nearest_neighbor_features_per_object (IntVOS.py:345-434) -> [h, w, n_ids]."""
    import ctypes
    lib = _lib.load()
    if _wants_grad(prev_frame_embedding, query_embedding):
        from .autograd import LocalMatchFn, LocalMatchFullFn
        prev = _hwc(prev_frame_embedding, "prev_frame_embedding")
        cur = _hwc(query_embedding, "query_embedding")
        if tuple(prev.shape) != tuple(cur.shape):
            raise ValueError("prev_frame_embedding and query_embedding must have the same shape")
        lab = _labels(prev_frame_labels, "prev_frame_labels")
        if lab.numel() != cur.shape[0] * cur.shape[1]:
            raise ValueError("prev_frame_labels must have height*width entries")
        return (LocalMatchFn if downsample else LocalMatchFullFn).apply(prev, cur, lab, n_ids, max_distance)
    both_bf16 = (prev_frame_embedding.dtype == torch.bfloat16 and query_embedding.dtype == torch.bfloat16
                 and downsample)  # 2-byte embeddings are read as they are by the pooling pass
    prev = _hwc(prev_frame_embedding, "prev_frame_embedding", both_bf16)
    cur = _hwc(query_embedding, "query_embedding", both_bf16)
    h, w, C = cur.shape
    if tuple(prev.shape) != (h, w, C):
        raise ValueError("prev_frame_embedding and query_embedding must have the same shape")
    lab = _labels(prev_frame_labels, "prev_frame_labels")
    if lab.numel() != h * w:
        raise ValueError("prev_frame_labels must have height*width entries")
    dev = cur.device
    nbytes = ctypes.c_size_t(0)
    _lib.check(lib.manet_local_workspace_bytes(h, w, C, max_distance, int(bool(downsample)),
                                               ctypes.byref(nbytes)), "manet_local_workspace_bytes")
    ws = _workspace(dev, "local", nbytes.value)
    out = torch.empty((h, w, n_ids), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = lib.manet_local_match_ex(prev.data_ptr(), prev.stride(0), prev.stride(1), prev.stride(2),
                                      cur.data_ptr(), cur.stride(0), cur.stride(1), cur.stride(2), _emb_code(cur),
                                      lab.data_ptr(), h, w, C, n_ids, max_distance, int(bool(downsample)),
                                      out.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "manet_local_match_ex")
    return out


def correlation_out_dims(H, W, pad_size, kernel_size, max_displacement, stride1, stride2):
    import ctypes
    lib = _lib.load()
    oc, oh, ow = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.manet_correlation_out_dims(H, W, pad_size, kernel_size, max_displacement, stride1, stride2,
                                              ctypes.byref(oc), ctypes.byref(oh), ctypes.byref(ow)),
               "manet_correlation_out_dims")
    return oc.value, oh.value, ow.value


def correlation_forward(input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2):
    """correlation_cuda.forward (correlation_cuda.cc:10-87) -> [B, (2r+1)^2, outH, outW] in the inputs' dtype
    (float32 / float16 / float64; anything else is computed in float32)."""
    lib = _lib.load()
    if _wants_grad(input1, input2):
        from .autograd import CorrelationFn
        _need_gpu(input1, "input1")
        _need_gpu(input2, "input2")
        return CorrelationFn.apply(input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2)
    _need_gpu(input1, "input1")
    _need_gpu(input2, "input2")
    # the reference's forward dispatches float / double / half (correlation_cuda_kernel.cu:386-415)
    code = {torch.float32: 0, torch.float16: 1, torch.float64: 2}.get(input1.dtype)
    if code is None or input2.dtype != input1.dtype:
        input1, input2, code = input1.float(), input2.float(), 0
    a = input1.contiguous()
    b = input2.contiguous()
    if a.dim() != 4 or a.shape != b.shape:
        raise ValueError("input1 and input2 must be [B, C, H, W] of the same shape")
    B, C, H, W = a.shape
    oc, oh, ow = correlation_out_dims(H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    out = torch.empty((B, oc, oh, ow), dtype=a.dtype, device=a.device)
    with _on(a.device):
        rc = lib.manet_correlation_forward(a.data_ptr(), b.data_ptr(), code, B, C, H, W, pad_size, kernel_size,
                                           max_displacement, stride1, stride2, out.data_ptr(),
                                           _stream_ptr(a.device))
    _lib.check(rc, "manet_correlation_forward")
    return out


def upsample_argmax(logits, size, want_small=True):
    """The mask step of the reference's driver in one launch:
    bilinear(align_corners) upsample of `logits` [1, n_ids, h, w] to `size`=(H, W) + argmax over ids
    (test.py:253-255) -> int64 [1, H, W]; and that mask nearest-resized back to (h, w) as the next
    frame's previous-frame label (IntVOS.py:598-599) -> int32 [1, 1, h, w] (None if not wanted)."""
    lib = _lib.load()
    _need_gpu(logits, "logits")
    # integer outputs (argmax): nothing to differentiate, same as the reference's torch.argmax
    if logits.dim() != 4 or logits.shape[0] != 1:
        raise ValueError("logits must be [1, n_ids, h, w]")
    lg = logits.float().contiguous()
    _, n_ids, h, w = lg.shape
    H, W = int(size[0]), int(size[1])
    mask = torch.empty((1, H, W), dtype=torch.int64, device=lg.device)
    small = torch.empty((1, 1, h, w), dtype=torch.int32, device=lg.device) if want_small else None
    with _on(lg.device):
        rc = lib.manet_upsample_argmax(lg.data_ptr(), n_ids, h, w, H, W, mask.data_ptr(),
                                       None if small is None else small.data_ptr(), _stream_ptr(lg.device))
    _lib.check(rc, "manet_upsample_argmax")
    return mask, small


def label_resize_nearest(mask, size):
    """F.interpolate(mask.float(), size=(h, w), mode='nearest').int() (IntVOS.py:598-599) in one launch.
    mask: integer [1, 1, H, W] (or [1, H, W] / [H, W]) -> int32 [1, 1, h, w]."""
    lib = _lib.load()
    _need_gpu(mask, "mask")
    if mask.is_floating_point() or mask.numel() != mask.shape[-1] * mask.shape[-2]:
        raise ValueError("mask must be ONE integer image [.., H, W]")
    m = mask.to(torch.int64).contiguous()
    H, W = int(m.shape[-2]), int(m.shape[-1])
    h, w = int(size[0]), int(size[1])
    out = torch.empty((1, 1, h, w), dtype=torch.int32, device=m.device)
    with _on(m.device):
        rc = lib.manet_label_resize_nearest(m.data_ptr(), H, W, h, w, out.data_ptr(), _stream_ptr(m.device))
    _lib.check(rc, "manet_label_resize_nearest")
    return out


def frame_begin(mask, size, fill=None, fill_value=1.0, scalar_dst=None, scalar_value=0.0):
    """label_resize_nearest(mask, size) plus up to two small device writes in the SAME launch (manet_frame_begin): `fill` (a
    contiguous float32 tensor) := fill_value -- the local map's slot pre-set for the wide windows -- and the one-element float32
    tensor `scalar_dst` := scalar_value -- the frame's distance weight in the caller's table (IntVOS.py:641).  Each was a ~5 us
    launch of its own in a propagated frame.  Returns the int32 [1, 1, h, w] label."""
    lib = _lib.load()
    _need_gpu(mask, "mask")
    if mask.is_floating_point() or mask.numel() != mask.shape[-1] * mask.shape[-2]:
        raise ValueError("mask must be ONE integer image [.., H, W]")
    m = mask.to(torch.int64).contiguous()
    H, W = int(m.shape[-2]), int(m.shape[-1])
    h, w = int(size[0]), int(size[1])
    for name, t, n in (("fill", fill, None), ("scalar_dst", scalar_dst, 1)):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.device != m.device or (n is not None and t.numel() != n)):
            raise ValueError("%s must be a contiguous float32 tensor on the mask's device%s" % (name, "" if n is None else " with one element"))
    out = torch.empty((1, 1, h, w), dtype=torch.int32, device=m.device)
    with _on(m.device):
        rc = lib.manet_frame_begin(m.data_ptr(), H, W, h, w, out.data_ptr(), None if fill is None else fill.data_ptr(),
                                   0 if fill is None else fill.numel(), float(fill_value),
                                   None if scalar_dst is None else scalar_dst.data_ptr(), float(scalar_value), _stream_ptr(m.device))
    _lib.check(rc, "manet_frame_begin")
    return out


def head_layer1_object(global_map, local_map, labels, n_ids, size, dw_weight, dw_bias, bn_scale, bn_shift, w2t_object, b2, term,
                       relu_out=True):
    """DynamicSegHead layer 1, per-object half, in one launch (manet_head_layer1_object_f32): the per-object input channels
    (IntVOS.py:663-669) -> depthwise 7x7 + bn1 + relu1 -> 1x1 (3 -> 256, bn2 folded) + `term` (the shared-embedding half,
    [1, 256, h, w]) [+ relu2] -> [n_ids, 256, h, w].  The same bits as head_inputs + dwconv7x7_bn_relu + conv1x1_mfma(add=term)."""
    _refuse_autograd("head_layer1_object", global_map, local_map, term)
    lib = _lib.load()
    _need_gpu(global_map, "global_map")
    h, w = int(size[0]), int(size[1])
    lab = labels.to(torch.int32).contiguous()
    g, l = global_map.contiguous(), local_map.contiguous()
    if g.dtype != torch.float32 or l.dtype != torch.float32 or g.numel() != h * w * n_ids or l.numel() != h * w * n_ids \
            or lab.numel() != h * w:
        raise ValueError("global_map / local_map must hold h*w*n_ids float32 values, labels h*w integers")
    wd = dw_weight.detach().float().contiguous()
    if wd.numel() != 3 * 49 or tuple(w2t_object.shape) != (3, PW_COUT) or term.numel() != PW_COUT * h * w:
        raise ValueError("the fused layer takes three per-object channels, 7x7 taps, %d output channels" % PW_COUT)
    opt = lambda t: None if t is None else t.detach().float().contiguous()  # noqa: E731
    db, sc, sh = opt(dw_bias), opt(bn_scale), opt(bn_shift)
    w2, bb, tm = w2t_object.contiguous(), b2.contiguous(), term.contiguous()
    for name, t in (("local_map", l), ("labels", lab), ("dw_weight", wd), ("dw_bias", db), ("bn_scale", sc), ("bn_shift", sh),
                    ("w2t_object", w2), ("b2", bb), ("term", tm)):
        if t is not None and t.device != g.device:
            raise ValueError("%s is on %s, global_map on %s" % (name, t.device, g.device))
    if w2.dtype != torch.float32 or bb.dtype != torch.float32 or tm.dtype != torch.float32 or bb.numel() != PW_COUT:
        raise ValueError("w2t_object [3, %d], b2 [%d] and term must be float32" % (PW_COUT, PW_COUT))
    if n_ids < 1:
        raise ValueError("n_ids must be positive")
    out = torch.empty((n_ids, PW_COUT, h, w), dtype=torch.float32, device=g.device)
    with _on(g.device):
        rc = lib.manet_head_layer1_object_f32(g.data_ptr(), l.data_ptr(), lab.data_ptr(), h, w, n_ids, wd.data_ptr(),
                                              None if db is None else db.data_ptr(), None if sc is None else sc.data_ptr(),
                                              None if sh is None else sh.data_ptr(), w2.data_ptr(), bb.data_ptr(), tm.data_ptr(),
                                              int(bool(relu_out)), out.data_ptr(), _stream_ptr(g.device))
    _lib.check(rc, "manet_head_layer1_object_f32")
    return out


def head_inputs(global_map, local_map, labels, n_ids, size):
    """The per-object channels of the head's input (IntVOS.py:663-669) in one launch:
    [n_ids, 3, h, w] = (global map of o, local map of o, labels == o), size = (h, w); global_map / local_map hold
    h*w*n_ids elements laid out [h, w, n_ids] (any view shape), labels h*w integers."""
    _refuse_autograd("head_inputs", global_map, local_map)
    lib = _lib.load()
    _need_gpu(global_map, "global_map")
    lab = labels.to(torch.int32).contiguous()
    h, w = int(size[0]), int(size[1])
    if lab.numel() != h * w or global_map.numel() != h * w * n_ids or local_map.numel() != h * w * n_ids:
        raise ValueError("global_map / local_map must hold h*w*n_ids elements as [h, w, n_ids], labels h*w")
    g, l = global_map.float().contiguous(), local_map.float().contiguous()
    out = torch.empty((n_ids, 3, h, w), dtype=torch.float32, device=g.device)
    with _on(g.device):
        rc = lib.manet_head_inputs_f32(g.data_ptr(), l.data_ptr(), lab.data_ptr(), h * w, n_ids, out.data_ptr(),
                                       _stream_ptr(g.device))
    _lib.check(rc, "manet_head_inputs_f32")
    return out


def fold_bn(bn):
    """eval-mode BatchNorm as per-channel (scale, shift): y = x * scale + shift"""
    inv = torch.rsqrt(bn.running_var.detach().float() + bn.eps)
    g = bn.weight.detach().float() if bn.weight is not None else torch.ones_like(inv)
    b = bn.bias.detach().float() if bn.bias is not None else torch.zeros_like(inv)
    scale = g * inv
    return scale, b - bn.running_mean.detach().float() * scale


def dwconv7x7_bn_relu(x, weight, bias=None, bn=None, relu=True, scale=None, shift=None, relu_in=False):
    """Depthwise 7x7 conv (padding 3) + bias + eval-mode BatchNorm + ReLU in one HIP kernel
    (IntVOS.py:491-493,500-502: conv1 -> bn1 -> relu1 of _split_separable_conv2d).
    x [B, C, h, w] fp32; weight [C, 1, 7, 7]; bn: an nn.BatchNorm2d in eval mode, or explicit per-channel
    `scale` / `shift` (fold_bn), or neither.  relu_in: read the input through max(x, 0) (the preceding block's
    relu2 folded into this pass)."""
    _refuse_autograd("dwconv7x7_bn_relu", x, weight, bias, scale, shift)  # callers use it under no_grad only
    lib = _lib.load()
    _need_gpu(x, "x")
    x = x.float().contiguous()
    B, C, h, w = x.shape
    if tuple(weight.shape) != (C, 1, 7, 7):
        raise ValueError("weight must be [C, 1, 7, 7]")
    wt = weight.detach().float().contiguous()
    if bn is not None:
        scale, shift = fold_bn(bn)
    if scale is not None:
        scale, shift = scale.float().contiguous(), shift.float().contiguous()
    bz = None if bias is None else bias.detach().float().contiguous()
    out = torch.empty_like(x)
    with _on(x.device):
        rc = lib.manet_dwconv7x7_bn_relu_ex(x.data_ptr(), B, C, h, w, wt.data_ptr(),
                                            None if bz is None else bz.data_ptr(),
                                            None if scale is None else scale.data_ptr(),
                                            None if shift is None else shift.data_ptr(), int(bool(relu)),
                                            int(bool(relu_in)), out.data_ptr(), _stream_ptr(x.device))
    _lib.check(rc, "manet_dwconv7x7_bn_relu_ex")
    return out


PW_COUT = 256  # output channels the MFMA 1x1 kernel is built for (the reference's MODEL_HEAD_EMBEDDING_DIM, config.py:48)


def fold_pointwise(conv2, bn2):
    """(w2t, b2) for conv1x1_mfma: conv2's [Cout, Cin, 1, 1] weight transposed to [Cin, Cout] with eval-mode bn2 folded
    in, and the folded bias [Cout]."""
    scale2, shift2 = fold_bn(bn2)
    w = conv2.weight.detach().float().reshape(conv2.out_channels, conv2.in_channels) * scale2[:, None]
    b2 = (conv2.bias.detach().float() * scale2 + shift2) if conv2.bias is not None else shift2
    return w.t().contiguous(), b2.contiguous()


def conv1x1_mfma_ok(x, cout):
    """shapes the MFMA 1x1 kernel takes: 256 output channels, any Cin, h*w a multiple of 4 (16-byte LDS-DMA rows)"""
    return (cout == PW_COUT and x.dim() == 4 and x.shape[1] >= 1 and (x.shape[2] * x.shape[3]) % 4 == 0
            and x.shape[2] * x.shape[3] >= 4)


def conv1x1_mfma(x, w2t, b2, relu_out=False, head_weight=None, head_bias=None, add=None):
    """conv2 -> bn2 [-> relu2] of a _split_separable_conv2d block (IntVOS.py:494,503-505) as an fp32-MFMA contraction fed by
    LDS-DMA (manet_conv1x1_f32).  x [B, Cin, h, w] fp32; w2t [Cin, 256], b2 [256] from fold_pointwise -> [B, 256, h, w].
    head_weight [1, 256, 1, 1] (+ head_bias [1]): DynamicSegHead's output layer Conv2d(256, 1, 1)(relu(.)) (IntVOS.py:519,525)
    fused into the epilogue -- returns [B, 1, h, w], the 256-channel activation is never written.
    add [256, h, w] / [1, 256, h, w] fp32: added to every batch entry before relu_out (layer1's shared-embedding half)."""
    _refuse_autograd("conv1x1_mfma", x, w2t, b2, head_weight, head_bias, add)
    lib = _lib.load()
    _need_gpu(x, "x")
    x = x.float().contiguous()
    B, cin, h, w = x.shape
    if tuple(w2t.shape) != (cin, PW_COUT) or b2.numel() != PW_COUT:
        raise ValueError("w2t must be [%d, %d] (ops.fold_pointwise), b2 [%d]" % (cin, PW_COUT, PW_COUT))
    if not conv1x1_mfma_ok(x, PW_COUT):
        raise ValueError("conv1x1_mfma needs h*w to be a multiple of 4")
    w2t, b2 = w2t.detach().float().contiguous(), b2.detach().float().contiguous()
    if add is not None:
        if head_weight is not None:
            raise ValueError("add and the fused output layer are exclusive")
        if add.numel() != PW_COUT * h * w or add.dtype != torch.float32 or add.device != x.device:
            raise ValueError("add must be [%d, h, w] fp32 on x's device" % PW_COUT)
        add = add.contiguous()
        out = torch.empty((B, PW_COUT, h, w), dtype=torch.float32, device=x.device)
        with _on(x.device):
            rc = lib.manet_conv1x1_add_f32(x.data_ptr(), cin * h * w, B, cin, h * w, w2t.data_ptr(), b2.data_ptr(),
                                           add.data_ptr(), PW_COUT, int(bool(relu_out)), out.data_ptr(), _stream_ptr(x.device))
        _lib.check(rc, "manet_conv1x1_add_f32")
        return out
    if head_weight is not None:
        if head_weight.numel() != PW_COUT:
            raise ValueError("head_weight must be [1, %d, 1, 1]" % PW_COUT)
        hw = head_weight.detach().float().contiguous()
        hb = None if head_bias is None else head_bias.detach().float().contiguous()
        hout = torch.empty((B, 1, h, w), dtype=torch.float32, device=x.device)
        with _on(x.device):
            rc = lib.manet_conv1x1_head_f32(x.data_ptr(), cin * h * w, B, cin, h * w, w2t.data_ptr(), b2.data_ptr(), PW_COUT,
                                            0, None, hw.data_ptr(), None if hb is None else hb.data_ptr(), hout.data_ptr(),
                                            _stream_ptr(x.device))
        _lib.check(rc, "manet_conv1x1_head_f32")
        return hout
    out = torch.empty((B, PW_COUT, h, w), dtype=torch.float32, device=x.device)
    with _on(x.device):
        rc = lib.manet_conv1x1_f32(x.data_ptr(), cin * h * w, B, cin, h * w, w2t.data_ptr(), b2.data_ptr(), PW_COUT,
                                   int(bool(relu_out)), out.data_ptr(), _stream_ptr(x.device))
    _lib.check(rc, "manet_conv1x1_f32")
    return out


class SplitWeight:
    """A 1x1 layer's folded weight packed for conv1x1_split (manet_conv1x1_x3_pack / _x6_pack): the MFMA A-operand image of
    its bf16 pieces -- pieces=2: hi + lo (16 significand bits, three products per pair), pieces=3: hi + mid + lo (24 bits,
    six products: fp32-class).  `cin` input channels, PW_COUT output channels."""

    def __init__(self, w2t, pieces=2):
        lib = _lib.load()
        _need_gpu(w2t, "w2t")
        if pieces not in (2, 3):
            raise ValueError("pieces must be 2 or 3")
        w2t = w2t.detach().float().contiguous()
        if w2t.dim() != 2 or w2t.shape[1] != PW_COUT:
            raise ValueError("w2t must be [Cin, %d] (ops.fold_pointwise)" % PW_COUT)
        self.cin, self.pieces = int(w2t.shape[0]), int(pieces)
        nbytes, pack = ((lib.manet_conv1x1_x6_weight_bytes, lib.manet_conv1x1_x6_pack) if pieces == 3
                        else (lib.manet_conv1x1_x3_weight_bytes, lib.manet_conv1x1_x3_pack))
        self.packed = torch.empty(int(nbytes(self.cin)), dtype=torch.uint8, device=w2t.device)
        with _on(w2t.device):
            rc = pack(w2t.data_ptr(), self.cin, PW_COUT, self.packed.data_ptr(), _stream_ptr(w2t.device))
        _lib.check(rc, "manet_conv1x1_x%d_pack" % (3 if pieces == 2 else 6))


def conv1x1_split_ok(x, cout):
    """shapes the split-bf16 1x1 kernel takes: 256 output channels, any Cin, h*w a multiple of 4"""
    return cout == PW_COUT and x.dim() == 4 and (x.shape[2] * x.shape[3]) % 4 == 0 and x.shape[2] * x.shape[3] >= 4


def conv1x1_split(x, weight, b2, relu_out=False, add=None, head_weight=None, head_bias=None):
    """conv2 -> bn2 [-> relu2] of a _split_separable_conv2d block (IntVOS.py:494,503-505) in split-bf16 arithmetic
    (manet_conv1x1_x3_f32: fp32 factors as hi + lo bf16 pieces, fp32 accumulation; <= 2^-16 relative per product; with a
    three-piece SplitWeight manet_conv1x1_x6_f32: hi + mid + lo, six products, ~2^-23 per product -- fp32-class).
    x [B, Cin, h, w] fp32; weight a SplitWeight; b2 [256] -> [B, 256, h, w].
    add [1, 256, h, w] (or [256, h, w]): added to every batch entry before relu_out -- layer1's shared-embedding half.
    head_weight / head_bias: DynamicSegHead's output layer fused (see conv1x1_mfma) -> [B, 1, h, w]."""
    _refuse_autograd("conv1x1_split", x, b2, add, head_weight, head_bias)
    lib = _lib.load()
    _need_gpu(x, "x")
    if not isinstance(weight, SplitWeight):
        raise TypeError("weight must be an ops.SplitWeight")
    x = x.float().contiguous()
    B, cin, h, w = x.shape
    if cin != weight.cin or b2.numel() != PW_COUT:
        raise ValueError("x has %d channels, the packed weight %d; b2 must be [%d]" % (cin, weight.cin, PW_COUT))
    if not conv1x1_split_ok(x, PW_COUT):
        raise ValueError("conv1x1_split needs h*w to be a multiple of 4")
    b2 = b2.detach().float().contiguous()
    if add is not None:
        if add.numel() != PW_COUT * h * w:
            raise ValueError("add must be [1, %d, h, w]" % PW_COUT)
        add = add.detach().float().contiguous()
    hw = hb = hout = out = None
    if head_weight is not None:
        if head_weight.numel() != PW_COUT:
            raise ValueError("head_weight must be [1, %d, 1, 1]" % PW_COUT)
        if add is not None:
            raise ValueError("add and the fused output layer are exclusive")
        hw = head_weight.detach().float().contiguous()
        hb = None if head_bias is None else head_bias.detach().float().contiguous()
        hout = torch.empty((B, 1, h, w), dtype=torch.float32, device=x.device)
    else:
        out = torch.empty((B, PW_COUT, h, w), dtype=torch.float32, device=x.device)
    ptr = lambda t: None if t is None else t.data_ptr()
    fn = lib.manet_conv1x1_x6_f32 if weight.pieces == 3 else lib.manet_conv1x1_x3_f32  # (the weight's packing decides)
    with _on(x.device):
        rc = fn(x.data_ptr(), cin * h * w, B, cin, h * w, weight.packed.data_ptr(), b2.data_ptr(),
                ptr(add), PW_COUT, int(bool(relu_out)), ptr(out), ptr(hw), ptr(hb), ptr(hout), _stream_ptr(x.device))
    _lib.check(rc, "manet_conv1x1_x6_f32" if weight.pieces == 3 else "manet_conv1x1_x3_f32")
    return hout if head_weight is not None else out


def relu_conv1x1_c1(x, weight, bias=None, relu_in=True):
    """DynamicSegHead's output layer in one pass over the activation (IntVOS.py:519,525): Conv2d(C, 1, 1) applied to
    max(x, 0) (relu_in) or to x.  x [B, C, h, w] fp32, weight [1, C, 1, 1], bias [1] or None -> [B, 1, h, w]."""
    _refuse_autograd("relu_conv1x1_c1", x, weight, bias)
    lib = _lib.load()
    _need_gpu(x, "x")
    x = x.float().contiguous()
    B, C, h, w = x.shape
    if weight.numel() != C:
        raise ValueError("weight must be [1, C, 1, 1]")
    wt = weight.detach().float().contiguous()
    bz = None if bias is None else bias.detach().float().contiguous()
    out = torch.empty((B, 1, h, w), dtype=torch.float32, device=x.device)
    with _on(x.device):
        rc = lib.manet_relu_conv1x1_c1_f32(x.data_ptr(), B, C, h * w, wt.data_ptr(), None if bz is None else bz.data_ptr(),
                                           int(bool(relu_in)), out.data_ptr(), _stream_ptr(x.device))
    _lib.check(rc, "manet_relu_conv1x1_c1_f32")
    return out
