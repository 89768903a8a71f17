"""ctypes binding of libmanet_hip.so -- the C ABI declared in include/manet_hip.h.

There is no fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmanet_hip.so")

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_i = ctypes.c_int
_sz = ctypes.c_size_t
_szp = ctypes.POINTER(ctypes.c_size_t)
_ip = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes); must list every symbol of include/manet_hip.h
SIGNATURES = {
    "manet_version": (ctypes.c_char_p, []),
    "manet_last_error_string": (ctypes.c_char_p, []),
    "manet_global_match_workspace_bytes": (_i, [_i64, _i64, _i, _i, _i, _i, _szp]),
    "manet_global_match": (_i, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _vp,
                                _vp, _i, _vp, _sz, _vp]),
    "manet_bank_workspace_bytes": (_i, [_i64, _i, _i, _i, _szp]),
    "manet_match_workspace_bytes": (_i, [_i64, _i64, _i, _i, _i, _i, _szp]),
    "manet_bank_prepare": (_i, [_vp, _i64, _i64, _vp, _i64, _i, _i, _i, _vp, _sz, _vp]),
    "manet_global_match_prepared": (_i, [_vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _vp, _vp, _i,
                                         _vp, _sz, _vp]),
    "manet_normalize_merge_f32": (_i, [_vp, _vp, _i64, _i, _vp]),
    "manet_local_workspace_bytes": (_i, [_i, _i, _i, _i, _i, _szp]),
    "manet_local_dist_f32": (_i, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i, _i, _i, _i, _i, _vp,
                                  _vp, _sz, _vp]),
    "manet_local_match_f32": (_i, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i, _i, _i, _i, _i,
                                   _i, _vp, _vp, _sz, _vp]),
    "manet_correlation_out_dims": (_i, [_i, _i, _i, _i, _i, _i, _i, _ip, _ip, _ip]),
    "manet_upsample_argmax": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "manet_dwconv7x7_bn_relu_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "manet_relu_conv1x1_c1_f32": (_i, [_vp, _i, _i, ctypes.c_long, _vp, _vp, _i, _vp, _vp]),
    "manet_dwconv7x7_bn_relu_ex": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "manet_profile_begin": (_i, [_i]),
    "manet_tune_set": (_i, [_i, _i]),
    "manet_profile_end": (_i, [ctypes.POINTER(ctypes.c_float), _i, _ip]),
    "manet_profile_end2": (_i, [ctypes.POINTER(ctypes.c_float), _i, _ip, ctypes.POINTER(ctypes.c_float), _i, _ip]),
    "manet_correlation_forward_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "manet_correlation_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "manet_bank_prepare_ex": (_i, [_vp, _i, _i64, _i64, _vp, _i64, _i, _i, _i, _vp, _sz, _vp]),
    "manet_query_pack_bytes": (_i, [_i64, _i, _i, _szp]),
    "manet_query_pack": (_i, [_vp, _i, _i64, _i64, _i64, _i, _i, _vp, _sz, _vp]),
    "manet_global_match_prepared_ex": (_i, [_vp, _i, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _vp, _vp, _i, _vp, _sz,
                                            _vp]),
    "manet_global_match_ex": (_i, [_vp, _i, _i64, _i64, _vp, _i, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _vp, _vp,
                                   _i, _vp, _sz, _vp]),
    "manet_local_match_ex": (_i, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i, _vp, _i, _i, _i, _i, _i, _i, _vp,
                                  _vp, _sz, _vp]),
    "manet_global_match_arg_workspace_bytes": (_i, [_i64, _i64, _i, _i, _szp]),
    "manet_head_layer1_object_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "manet_local_match_full_arg_f32": (_i, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp,
                                            _vp]),
    "manet_local_match_full_backward_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "manet_global_match_topk_arg_workspace_bytes": (_i, [_i64, _i64, _i, _i, _szp]),
    "manet_global_match_topk_arg_f32": (_i, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _vp, _vp, _vp, _sz,
                                             _vp]),
    "manet_global_match_arg_f32": (_i, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _vp, _vp, _vp, _sz,
                                        _vp]),
    "manet_global_match_backward_f32": (_i, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _i, _i, _vp,
                                             _i64, _i64, _vp, _i64, _i64, _vp]),
    "manet_local_match_arg_workspace_bytes": (_i, [_i, _i, _i, _i, _szp]),
    "manet_local_match_arg_f32": (_i, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i, _i, _i, _i, _i, _vp,
                                       _vp, _vp, _vp, _sz, _vp]),
    "manet_local_match_backward_workspace_bytes": (_i, [_i, _i, _i, _i, _szp]),
    "manet_local_match_backward_f32": (_i, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i, _i, _i,
                                            _i, _i, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _sz, _vp]),
    "manet_conv1x1_f32": (_i, [_vp, _i64, _i, _i, _i64, _vp, _vp, _i, _i, _vp, _vp]),
    "manet_conv1x1_head_f32": (_i, [_vp, _i64, _i, _i, _i64, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "manet_conv1x1_add_f32": (_i, [_vp, _i64, _i, _i, _i64, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "manet_label_resize_nearest": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "manet_frame_begin": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i64, ctypes.c_float, _vp, ctypes.c_float, _vp]),
    "manet_head_inputs_f32": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "manet_conv1x1_x3_weight_bytes": (_i64, [_i]),
    "manet_conv1x1_x3_pack": (_i, [_vp, _i, _i, _vp, _vp]),
    "manet_conv1x1_x3_f32": (_i, [_vp, _i64, _i, _i, _i64, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "manet_conv1x1_x6_weight_bytes": (_i64, [_i]),
    "manet_conv1x1_x6_pack": (_i, [_vp, _i, _i, _vp, _vp]),
    "manet_conv1x1_x6_f32": (_i, [_vp, _i64, _i, _i, _i64, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "manet_global_match_refine": (_i, [_vp, _i, _i64, _i64, _vp, _vp, _i64, _i64, _i, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "manet_global_match_refine_stats": (_i, [_vp, _i64, _i, _i, ctypes.POINTER(ctypes.c_int64),
                                             ctypes.POINTER(ctypes.c_int64)]),
    "manet_global_match_refine_stats2": (_i, [_vp, _i64, _i, _i, ctypes.POINTER(ctypes.c_int64)]),
    "manet_global_match_refine_rescued_async": (_i, [_vp, _i64, _i, _i, _vp, _vp]),
    "manet_frame_workspace_bytes": (_i, [_i, _i, _i, _i, _i, _szp]),
    "manet_frame_prepare": (_i, [_vp, _i, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _i64,
                                 ctypes.c_uint32, _vp]),
    "manet_embed_finish": (_i, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "manet_local_match_frames": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "manet_local_volume_bytes": (_i, [_i, _i, _i, ctypes.POINTER(ctypes.c_size_t)]),
    "manet_local_volume_frames": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "manet_local_match_volume": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "manet_profile_read": (_i, [_i, ctypes.POINTER(ctypes.c_float), _i, _ip]),
    "manet_correlation_backward_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "manet_correlation_backward_f64": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
}

COMPUTE_F32, COMPUTE_BF16, COMPUTE_BF16X3, COMPUTE_BF16_REFINE = 0, 1, 2, 3
EMB_F32, EMB_BF16, EMB_PACKED = 0, 1, 2
EPI_NORMALIZE = 1
EPI_KEYS_ARMED = 2
EPI_REFINE_EXACT = 4

_lib = None


def load():
    """Load libmanet_hip.so (built by csrc/Makefile, see __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "cvpr2020_manet_amd: %s is missing -- build it with "
                "`make -C cvpr2020_manet_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().manet_last_error_string().decode("utf-8", "replace")
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, msg))
