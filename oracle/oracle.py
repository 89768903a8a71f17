"""CPU ORACLE -- test infrastructure, NOT product code.

ctypes front-end of ``oracle/manet_oracle.c`` (a plain-C restatement of the reference's
matching path, ``/root/reference/networks/IntVOS.py:23-434`` plus
``correlation_package/correlation_cuda_kernel.cu:46-147``) and a few numpy-only helpers for the
pieces of the path that are pure indexing (local-map memory select, ``IntVOS.py:638-661``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; nothing under ``cvpr2020_manet_amd/`` does.

Parity pinning: see the header of ``manet_oracle.c`` -- pinned against golden vectors generated
from the imported reference by ``oracle/gen_golden.py`` (``tests/golden/*.npz``).

All array arguments are numpy arrays (any strides; float32 / int32); results are fresh arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmanet_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_L = ctypes.c_long
_I = ctypes.c_int


def build(force=False):
    """Compile the C oracle with the committed Makefile (gcc only, seconds)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "manet_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.oracle_global_match_f32.restype = _I
        L.oracle_global_match_f32.argtypes = [_f32p, _L, _L, _f32p, _L, _L, _i32p, _L, _L, _I, _I,
                                              _I, _I, _I, _f32p]
        L.oracle_normalize_merge_f32.restype = None
        L.oracle_normalize_merge_f32.argtypes = [_f32p, _f32p, _L, _I]
        L.oracle_local_dist_f32.restype = None
        L.oracle_local_dist_f32.argtypes = [_f32p, _L, _L, _L, _f32p, _L, _L, _L, _I, _I, _I, _I, _I,
                                            _f32p]
        L.oracle_local_masked_min_f32.restype = None
        L.oracle_local_masked_min_f32.argtypes = [_f32p, _i32p, _I, _I, _I, _I, _f32p]
        L.oracle_local_match_f32.restype = None
        L.oracle_local_match_f32.argtypes = [_f32p, _L, _L, _L, _f32p, _L, _L, _L, _i32p, _I, _I, _I,
                                             _I, _I, _I, _f32p]
        L.oracle_correlation_out_dims.restype = _I
        L.oracle_correlation_out_dims.argtypes = [_I] * 7 + [ctypes.POINTER(_I)] * 3
        L.oracle_correlation_forward_f32.restype = _I
        L.oracle_correlation_forward_f32.argtypes = [_f32p, _f32p] + [_I] * 9 + [_f32p]
        L.oracle_upsample_argmax.restype = None
        L.oracle_upsample_argmax.argtypes = [_f32p, _I, _I, _I, _I, _I, ctypes.POINTER(ctypes.c_int64), _i32p]
        L.oracle_dwconv7x7_bn_relu_f32.restype = None
        L.oracle_dwconv7x7_bn_relu_f32.argtypes = [_f32p, _I, _I, _I, _I, _f32p, _f32p, _f32p, _f32p, _I, _f32p]
        L.oracle_bf16_round.restype = ctypes.c_float
        L.oracle_bf16_round.argtypes = [ctypes.c_float]
        _lib = L
    return _lib


def _f32(a):
    a = np.asarray(a)
    if a.dtype != np.float32:
        a = a.astype(np.float32)
    return a


def _ptr(a):
    return a.ctypes.data_as(_f32p)


def _estr(a):
    """element strides of a float32 array"""
    return [s // 4 for s in a.strides]


def _flat2d(a):
    """[..., C] -> strides of the flattened [rows, C] view (must be viewable, as torch .view)."""
    C = a.shape[-1]
    rows = int(np.prod(a.shape[:-1]))
    st = _estr(a)
    # check the leading dims collapse into one stride (torch .view(-1, C) semantics)
    lead_shape, lead_st = list(a.shape[:-1]), st[:-1]
    while len(lead_shape) > 1:
        if lead_shape[-1] == 1:      # a size-1 dim carries no stride information
            merged = lead_st[-2]
        elif lead_shape[-2] == 1:
            merged = lead_st[-1]
        elif lead_st[-2] == lead_st[-1] * lead_shape[-1]:
            merged = lead_st[-1]
        else:
            raise ValueError("leading dims are not collapsible; pass a .view(-1, C)-able array")
        lead_st[-2:] = [merged]
        lead_shape[-2:] = [lead_shape[-2] * lead_shape[-1]]
    return rows, C, lead_st[0], st[-1]


def global_match(reference_embeddings, query_embeddings, reference_labels, k_nearest_neighbors=1,
                 n_ids=None, test_mode=True, quant_bf16=False):
    """nearest_neighbor_features_per_object (IntVOS.py:160-210).

    reference_embeddings [h_r, w_r, C], query_embeddings [h, w, C], reference_labels
    [h_r, w_r, 1] int.  n_ids = gt_ids + 1 (None: max label + 1, IntVOS.py:192-198).
    Returns float32 [1, h, w, n_ids, 1] of raw (un-normalised) distances.
    """
    ref = _f32(reference_embeddings)
    qry = _f32(query_embeddings)
    lab = np.ascontiguousarray(np.asarray(reference_labels).reshape(-1).astype(np.int32))
    assert ref.shape[:2] == tuple(np.asarray(reference_labels).shape[:2])  # IntVOS.py:189
    h, w, _ = qry.shape
    if n_ids is None:
        n_ids = int(lab.max()) + 1
    M0, C, bs_m, bs_c = _flat2d(ref)
    N, C2, qs_n, qs_c = _flat2d(qry)
    assert C == C2
    out = np.empty((N, n_ids), np.float32)
    rc = lib().oracle_global_match_f32(_ptr(qry), qs_n, qs_c, _ptr(ref), bs_m, bs_c,
                                       lab.ctypes.data_as(_i32p), N, M0, C, n_ids,
                                       int(k_nearest_neighbors), int(bool(test_mode)),
                                       int(bool(quant_bf16)), _ptr(out))
    if rc != 0:
        raise RuntimeError("selected index k out of range")  # what torch.topk raises
    return out.reshape(1, h, w, n_ids, 1)


def normalize_merge(x, mem=None, normalize=True):
    """IntVOS.py:611-612 and :620-622.  Returns (g, new_mem) as new arrays."""
    g = np.ascontiguousarray(_f32(x)).copy()
    m = None if mem is None else np.ascontiguousarray(_f32(mem)).copy()
    lib().oracle_normalize_merge_f32(_ptr(g), None if m is None else _ptr(m), g.size,
                                     int(bool(normalize)))
    return g, m


def local_dist(x, y, max_distance, downsample=True):
    """local_pairwise_distances2(x=query, y=prev) (IntVOS.py:266-315) -> [h, w, (2d+1)^2]."""
    x = _f32(x)
    y = _f32(y)
    h, w, C = x.shape
    P = 2 * max_distance + 1
    out = np.empty((h, w, P * P), np.float32)
    xs, ys = _estr(x), _estr(y)
    lib().oracle_local_dist_f32(_ptr(x), xs[0], xs[1], xs[2], _ptr(y), ys[0], ys[1], ys[2], h, w, C,
                                int(max_distance), int(bool(downsample)), _ptr(out))
    return out


def local_masked_min(dist, labels, max_distance, n_ids):
    """IntVOS.py:398-408,428-432 -> [1, h, w, n_ids, 1]."""
    dist = np.ascontiguousarray(_f32(dist))
    h, w, _ = dist.shape
    lab = np.ascontiguousarray(np.asarray(labels).reshape(h, w).astype(np.int32))
    out = np.empty((h, w, n_ids), np.float32)
    lib().oracle_local_masked_min_f32(_ptr(dist), lab.ctypes.data_as(_i32p), h, w,
                                      int(max_distance), int(n_ids), _ptr(out))
    return out.reshape(1, h, w, n_ids, 1)


def local_match(prev_frame_embedding, query_embedding, prev_frame_labels, n_ids, max_distance=12,
                downsample=True):
    """local_previous_frame_nearest_neighbor_features_per_object (IntVOS.py:345-434)."""
    prev = _f32(prev_frame_embedding)
    cur = _f32(query_embedding)
    h, w, C = cur.shape
    lab = np.ascontiguousarray(np.asarray(prev_frame_labels).reshape(h, w).astype(np.int32))
    out = np.empty((h, w, n_ids), np.float32)
    ps, cs = _estr(prev), _estr(cur)
    lib().oracle_local_match_f32(_ptr(prev), ps[0], ps[1], ps[2], _ptr(cur), cs[0], cs[1], cs[2],
                                 lab.ctypes.data_as(_i32p), h, w, C, int(max_distance), int(n_ids),
                                 int(bool(downsample)), _ptr(out))
    return out.reshape(1, h, w, n_ids, 1)


def correlation_out_dims(H, W, pad_size, kernel_size, max_displacement, stride1, stride2):
    oc, oh, ow = _I(), _I(), _I()
    rc = lib().oracle_correlation_out_dims(H, W, pad_size, kernel_size, max_displacement, stride1,
                                           stride2, ctypes.byref(oc), ctypes.byref(oh),
                                           ctypes.byref(ow))
    if rc != 0:
        raise ValueError("empty correlation output")
    return oc.value, oh.value, ow.value


def correlation_forward(input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2):
    """correlation_cuda.forward (correlation_cuda.cc:10-87) -> [B, (2r+1)^2, outH, outW]."""
    a = np.ascontiguousarray(_f32(input1))
    b = np.ascontiguousarray(_f32(input2))
    B, C, H, W = a.shape
    oc, oh, ow = correlation_out_dims(H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    out = np.empty((B, oc, oh, ow), np.float32)
    rc = lib().oracle_correlation_forward_f32(_ptr(a), _ptr(b), B, C, H, W, pad_size, kernel_size,
                                              max_displacement, stride1, stride2, _ptr(out))
    assert rc == 0
    return out


def bf16_round(a):
    """round-to-nearest-even to bf16, returned as float32 (numpy, vectorised)."""
    a = np.ascontiguousarray(_f32(a))
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).reshape(a.shape)


def local_map_select(local_map_tmp, local_map_dist, new_map, frame, interaction_num,
                     start_annotated_frame):
    """Local-map memory update and selection (IntVOS.py:638-661), numpy, in place.

    local_map_tmp [104, 9, h, w, n_ids, 1], local_map_dist [104, 9]; new_map [1, h, w, n_ids, 1].
    Returns the map that the head consumes, [1, h, w, n_ids, 1].
    """
    local_map_dist[frame][interaction_num - 1] = 1.0 / abs(frame - start_annotated_frame)
    local_map_tmp[frame][interaction_num - 1] = new_map[0]
    if interaction_num == 1:
        sel = local_map_tmp[frame][interaction_num - 1]
    elif local_map_dist[frame][interaction_num - 1] > local_map_dist[frame][interaction_num - 2]:
        sel = local_map_tmp[frame][interaction_num - 1]
    else:
        sel = local_map_tmp[frame][interaction_num - 2]
    return sel[None]


def upsample_argmax(logits, size):
    """test.py:253-255 + IntVOS.py:598-599: logits [1, n_ids, h, w] -> (mask int64 [1,H,W], small int32 [1,1,h,w])."""
    lg = np.ascontiguousarray(_f32(logits))
    _, n_ids, h, w = lg.shape
    H, W = int(size[0]), int(size[1])
    mask = np.empty((H, W), np.int64)
    small = np.empty((h, w), np.int32)
    lib().oracle_upsample_argmax(_ptr(lg), n_ids, h, w, H, W, mask.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                 small.ctypes.data_as(_i32p))
    return mask[None], small[None, None]


def dwconv7x7_bn_relu(x, weight, bias=None, scale=None, shift=None, relu=True):
    """IntVOS.py:491-493,500-502 (conv1 -> bn1 -> relu1), BN folded to per-channel scale/shift."""
    x = np.ascontiguousarray(_f32(x))
    B, C, h, w = x.shape
    wt = np.ascontiguousarray(_f32(weight)).reshape(C, 49)
    opt = lambda a: None if a is None else _ptr(np.ascontiguousarray(_f32(a)))
    keep = [np.ascontiguousarray(_f32(a)) if a is not None else None for a in (bias, scale, shift)]
    out = np.empty_like(x)
    lib().oracle_dwconv7x7_bn_relu_f32(_ptr(x), B, C, h, w, _ptr(wt), *[None if k is None else _ptr(k) for k in keep],
                                       int(bool(relu)), _ptr(out))
    return out


def global_match_blas(reference_embeddings, query_embeddings, reference_labels, n_ids, chunk=1024):
    """The reference's own formulation of the k = 1 global match on the host's BLAS: per query chunk the dense product
    x @ y^T (torch's CPU GEMM, all cores), d = (xs + ys) - 2 mm (IntVOS.py:32-39), then the masked minimum per object
    (IntVOS.py:81-85) -- on a bank pre-sorted by object id, so the mask is a column slice.  The arithmetic ORDER of the GEMM is the
    library's, not the reference chain's: this is the CPU BASELINE's fast form (bench.py `cpu_baseline`, kind "port": a BLAS-shaped
    port), not the parity oracle -- results agree with `global_match` to accumulation-order rounding.
    reference_embeddings [M, C], query_embeddings [N, C] float32, reference_labels [M] int -> float32 [N, n_ids]."""
    import torch
    ref = torch.from_numpy(np.ascontiguousarray(_f32(reference_embeddings).reshape(-1, reference_embeddings.shape[-1])))
    qry = torch.from_numpy(np.ascontiguousarray(_f32(query_embeddings).reshape(-1, query_embeddings.shape[-1])))
    lab = torch.from_numpy(np.ascontiguousarray(reference_labels).reshape(-1).astype(np.int64))
    order = torch.argsort(lab, stable=True)
    lab_s, ref_s = lab[order], ref[order].contiguous()
    bounds = torch.searchsorted(lab_s, torch.arange(0, n_ids + 1))
    ys = (ref_s * ref_s).sum(1)[None, :]
    out = torch.full((qry.shape[0], n_ids), 1e20, dtype=torch.float32)
    ref_t = ref_s.t().contiguous()
    for i0 in range(0, qry.shape[0], chunk):
        x = qry[i0:i0 + chunk]
        d = (x * x).sum(1, keepdim=True) + ys
        d.addmm_(x, ref_t, beta=1.0, alpha=-2.0)
        for o in range(n_ids):
            a, b = int(bounds[o]), int(bounds[o + 1])
            if b > a:
                out[i0:i0 + chunk, o] = d[:, a:b].min(1).values
    return out.numpy()
