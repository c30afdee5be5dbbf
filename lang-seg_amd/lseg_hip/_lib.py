"""ctypes binding of liblseg_hip.so (C ABI declared in include/lseg_hip.h).

The library is built in-tree (lang-seg_amd/csrc/Makefile -> lseg_hip/liblseg_hip.so).
There is no fallback: if the shared object is missing `load()` raises, and on a
box without a gfx950 GPU every compute entry point returns LSEG_ERR_NO_DEVICE.
"""
import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSEG_HIP_LIB", os.path.join(_HERE, "liblseg_hip.so"))   # override: A/B-testing builds

LSEG_F32, LSEG_F16, LSEG_BF16, LSEG_I64, LSEG_F16_SPLIT = 0, 1, 2, 3, 4
RS_IDENTITY, RS_CONVT, RS_CONV_S2 = 0, 1, 2
ABI_VERSION = 1

STATUS = {0: "LSEG_OK", -1: "LSEG_ERR_INVALID", -2: "LSEG_ERR_NO_DEVICE", -3: "LSEG_ERR_HIP",
          -4: "LSEG_ERR_STATE", -5: "LSEG_ERR_UNSUPPORTED", -6: "LSEG_ERR_MISSING_PARAM"}


class LSegConfigC(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("patch", C.c_int32), ("dim", C.c_int32), ("depth", C.c_int32), ("heads", C.c_int32),
        ("hooks", C.c_int32 * 4), ("pos_grid", C.c_int32),
        ("reassemble_ch", C.c_int32 * 4), ("resample_kind", C.c_int32 * 4), ("resample_k", C.c_int32 * 4),
        ("features", C.c_int32), ("out_c", C.c_int32),
        ("arch_option", C.c_int32), ("block_depth", C.c_int32), ("activation", C.c_int32),
        ("text_vocab", C.c_int32), ("text_ctx", C.c_int32), ("text_width", C.c_int32),
        ("text_heads", C.c_int32), ("text_layers", C.c_int32),
        ("img_h", C.c_int32), ("img_w", C.c_int32),
        ("max_batch", C.c_int32), ("max_labels", C.c_int32),
        ("image_dtype", C.c_int32), ("flags", C.c_int32),
    ]


class LSegError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


# every exported symbol of include/lseg_hip.h: name -> (restype, argtypes)
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
SIGNATURES = {
    "lseg_abi_version": (_i, []),
    "lseg_device_count": (_i, []),
    "lseg_create": (_i, [C.POINTER(LSegConfigC), _i, C.POINTER(_vp)]),
    "lseg_destroy": (_i, [_vp]),
    "lseg_last_error": (C.c_char_p, [_vp]),
    "lseg_bind_param": (_i, [_vp, C.c_char_p, _vp, _i, C.POINTER(C.c_int64), _i]),
    "lseg_finalize_params": (_i, [_vp, _vp]),
    "lseg_set_text_tokens": (_i, [_vp, C.POINTER(C.c_int64), _i, _i]),
    "lseg_set_text_features": (_i, [_vp, _vp, _i, _vp]),
    "lseg_encode_text": (_i, [_vp, _vp]),
    "lseg_set_text_cache": (_i, [_vp, _i]),
    "lseg_get_text_features": (_i, [_vp, _vp, _vp]),
    "lseg_set_text_grouping": (_i, [_vp, _i]),
    "lseg_overflow_seen": (_i, [_vp, _i]),
    "lseg_forward": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "lseg_forward_stats": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "lseg_get_intermediate": (_i, [_vp, C.c_char_p, _vp, _sz, C.POINTER(_sz), _vp]),
    "lseg_set_debug": (_i, [_vp, _i]),
    "lseg_set_profiling": (_i, [_vp, _i]),
    "lseg_check_range": (_i, [_vp, C.POINTER(C.c_uint64), _vp]),
    "lseg_get_profile": (_i, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "lseg_set_train": (_i, [_vp, _i]),
    "lseg_bind_grad": (_i, [_vp, C.c_char_p, _vp]),
    "lseg_grad_ptr": (_i, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_sz)]),
    "lseg_grad_bucket": (_i, [_vp, C.c_char_p]),
    "lseg_num_grad_buckets": (_i, [_vp]),
    "lseg_backward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "lseg_backward_scaled": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "lseg_train_loss": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "lseg_sgd_momentum": (_i, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_sz)]),
    "lseg_sgd_mark_initialized": (_i, [_vp, _i]),
    "lseg_set_bn_sync": (_i, [_vp, _vp, _vp, _i]),
    "lseg_set_bucket_callback": (_i, [_vp, _vp, _vp]),
    "lseg_sgd_step": (_i, [_vp, _f, _f, _f, _f, _vp]),
    "lseg_op_corr_planes": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "lseg_op_gemm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "lseg_op_gemm_vit": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "lseg_op_gemm_res32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "lseg_op_colsum": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, C.c_size_t, _vp]),
    "lseg_op_bn_stats": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, C.c_size_t, _vp]),
    "lseg_op_bn_bwd_stats": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, C.c_size_t, _vp]),
    "lseg_op_layernorm": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "lseg_op_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "lseg_op_attention_prescaled": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "lseg_op_conv3x3": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "lseg_op_upsample2x_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "lseg_op_upsample2x_planes": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "lseg_op_upsample4x_planes_scaled": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lseg_op_correlation": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "lseg_op_head_features": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "lseg_op_seg_stats": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "lseg_op_seg_stats_lowres": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "lseg_op_linear_backward": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "lseg_op_attention_backward_ws_bytes": (_sz, [_i, _i, _i]),
    "lseg_op_attention_backward_qkv": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "lseg_op_quickgelu_backward": (_i, [_vp, _vp, _vp, C.c_int64, _i, _vp]),
    "lseg_op_bn_train_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "lseg_op_bn_train_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "lseg_op_relu_backward": (_i, [_vp, _vp, _vp, C.c_int64, _vp]),
    "lseg_op_upsample2x_planes_backward_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "lseg_op_upsample_ce_backward_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "lseg_op_eval_make_crops": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "lseg_op_eval_accumulate": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "lseg_op_eval_resize": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "lseg_op_l2norm_scale_backward": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _f, _vp]),
    "lseg_op_gelu_backward": (_i, [_vp, _vp, _vp, C.c_int64, _i, _vp]),
    "lseg_op_upsample2x_nhwc_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "lseg_op_softmax_ce_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lseg_op_conv3x3_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "lseg_op_layernorm_backward": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp]),
}

# callback types of the training step (include/lseg_hip.h: lseg_reduce_cb, lseg_bucket_cb)
REDUCE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
BUCKET_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the engine; raises (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP engine first "
            f"(python -c 'import __graft_entry__ as g; g.build()' or make -C lang-seg_amd/csrc). "
            f"There is no CPU/PyTorch fallback for the LSeg forward path.")
    lib = C.CDLL(LIB_PATH)
    # tools only: an OLDER build loaded through LSEG_HIP_LIB for a same-box A/B may lack entry points added since (LSEG_HIP_ALLOW_MISSING=1)
    lenient = "LSEG_HIP_LIB" in os.environ and os.environ.get("LSEG_HIP_ALLOW_MISSING", "") == "1"
    for name, (res, args) in SIGNATURES.items():
        if lenient and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.lseg_abi_version() != ABI_VERSION:
        raise ImportError(f"liblseg_hip.so ABI {lib.lseg_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int):
    if code != 0:
        msg = load().lseg_last_error(None)
        raise LSegError(code, msg.decode("utf-8", "replace") if msg else "")
