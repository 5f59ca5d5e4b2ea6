"""ctypes binding of libsdhip.so (the C ABI declared in include/sd_hip.h).

There is NO fallback: if the library is missing or a symbol is absent this module raises, and
every product path above it fails loudly.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SD_AMD_LIB", os.path.join(HERE, "lib", "libsdhip.so"))   # (SD_AMD_LIB: A/B builds of the kernels)
HEADER_PATH = os.path.normpath(os.path.join(HERE, "..", "include", "sd_hip.h"))

# the SD_ABLATE build of the same sources (build.py): timing-ablation kernels that produce WRONG results by design
# (sd_op_conv3x3_ablate's DIAG modes, SD_GEMM_TUNE) live ONLY there -- tools/, bench.py's MFMA-stream probe and one test load it
ABLATE_LIB_PATH = os.path.join(HERE, "lib", "libsdhip_ablate.so")

_lib: Optional[C.CDLL] = None
_ablate: Optional[C.CDLL] = None


class SdUnetConfig(C.Structure):
    _fields_ = [
        ("sample_size", C.c_int), ("in_channels", C.c_int), ("out_channels", C.c_int),
        ("num_levels", C.c_int), ("block_out_channels", C.c_int * 8), ("layers_per_block", C.c_int),
        ("attn_levels", C.c_int * 8), ("cross_attention_dim", C.c_int), ("num_heads", C.c_int),
        ("norm_num_groups", C.c_int), ("norm_eps", C.c_float), ("context_len", C.c_int),
        ("weight_dtype", C.c_int), ("fp8_act_scale_norm", C.c_float), ("fp8_act_scale_ff", C.c_float),
    ]


DTYPE_BF16, DTYPE_FP8_E4M3 = 0, 1
DTYPES = {"bf16": DTYPE_BF16, "fp8": DTYPE_FP8_E4M3, "fp8_e4m3": DTYPE_FP8_E4M3}


class SdClipConfig(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int), ("hidden_size", C.c_int), ("num_layers", C.c_int), ("num_heads", C.c_int),
        ("intermediate_size", C.c_int), ("max_positions", C.c_int), ("layer_norm_eps", C.c_float),
    ]


class SdHipError(RuntimeError):
    pass


def declared_symbols() -> list:
    """Every function name include/sd_hip.h declares (used by the CPU export test)."""
    src = open(HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sd_[a-z0-9_]+)\s*\(", src)))


_vp, _i, _ll, _f = C.c_void_p, C.c_int, C.c_longlong, C.c_float
_SIGS = {
    "sd_last_error": (C.c_char_p, []),
    "sd_abi_version": (_i, []),
    "sd_unet_create": (_i, [C.POINTER(SdUnetConfig), C.POINTER(_vp)]),
    "sd_unet_destroy": (None, [_vp]),
    "sd_unet_num_params": (_i, [_vp]),
    "sd_unet_param_info": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(_ll), C.POINTER(_i)]),
    "sd_unet_load_param": (_i, [_vp, C.c_char_p, _vp, _ll]),
    "sd_unet_finalize": (_i, [_vp]),
    "sd_unet_debug_packed": (_ll, [_vp, C.c_char_p, _vp, _ll]),
    "sd_unet_calibrate_fp8": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _vp, _ll]),
    "sd_unet_fp8_scale_count": (_i, [_vp]),
    "sd_unet_fp8_scale_info": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(_f), C.POINTER(_f)]),
    "sd_unet_set_fp8_scale": (_i, [_vp, C.c_char_p, _f]),
    "sd_unet_workspace_bytes": (_ll, [_vp, _i, _i]),
    "sd_unet_set_context": (_i, [_vp, _vp, _vp, _i, _i, _vp, _ll]),
    "sd_unet_forward": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _ll, _i, _i]),
    "sd_unet_forward_profiled": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _ll, _i, _i, C.POINTER(C.c_double),
                                      C.POINTER(_ll), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sd_unet_forward_op_times": (_ll, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _ll, _i, _i, C.c_char_p, _ll]),
    "sd_vae_create": (_i, [C.POINTER(SdUnetConfig), C.POINTER(_vp)]),
    "sd_vae_decode": (_i, [_vp, _vp, _vp, _i, _f, _vp, _vp, _ll]),
    "sd_clip_create": (_i, [C.POINTER(SdClipConfig), C.POINTER(_vp)]),
    "sd_clip_encode": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _ll]),
    "sd_unet_debug_tensor": (_i, [_vp, _vp, C.c_char_p, _vp, _ll, _vp, _i, _i]),
    "sd_sched_step": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_f), _ll]),
    "sd_op_gemm": (_i, [_vp, _vp, _ll, _vp, _ll, _i, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _i, _i, _i, _i]),
    "sd_op_gemm_batched": (_i, [_vp, _vp, _ll, _vp, _ll, _i, _vp, _vp, _ll, _vp, _ll, _i, _i, _i, _i, _i]),
    "sd_op_gemm_batched_softmax_ln": (_i, [_vp, _vp, _ll, _vp, _ll, _i, _vp, _ll, _i, _i, _i, _i, _vp, _i, _vp, _vp, _f]),
    "sd_op_conv3x3": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    "sd_op_conv3x3_ablate": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "sd_op_conv3x3_upsample_subpixel": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "sd_op_conv3x3_upsample_subpixel_groupnorm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _f, _i]),
    "sd_op_groupnorm": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _i]),
    "sd_op_conv3x3_groupnorm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _f, _i]),
    "sd_op_conv3x3_splitk": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "sd_op_layernorm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f]),
    "sd_op_attention": (_i, [_vp, _vp, _ll, _vp, _ll, _vp, _ll, _vp, _ll, _i, _i, _i, _i, _i, _f]),
    "sd_op_gemm_qkv_headmajor": (_i, [_vp, _vp, _ll, _vp, _vp, _vp, _i, _i, _i, _i]),
    "sd_op_attention_headmajor": (_i, [_vp, _vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _f]),
    "sd_op_clip_attention": (_i, [_vp, _vp, _vp, _i, _i, _i, _i]),
    "sd_op_conv_in": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "sd_op_conv_out": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "sd_op_time_embedding": (_i, [_vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i]),
    "sd_op_ln_partials": (_i, [_i, _i, _i]),
    "sd_op_gemm_rowstats": (_i, [_vp, _vp, _ll, _vp, _vp, _vp, _ll, _vp, _ll, _i, _i, _i, _vp]),
    "sd_op_gemm_plan": (_i, [_vp, _vp, _ll, _vp, _ll, _i, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _f,
                            _vp, _i]),
    "sd_op_gemm_ln": (_i, [_vp, _vp, _ll, _vp, _vp, _vp, _vp, _i, _f, _vp, _ll, _i, _i, _i, _i]),
    "sd_op_xattn_fused_rowstats": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sd_op_xattn_fused": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "sd_op_xattn_fused_ln": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _ll, _vp, _f, _vp]),
    "sd_op_xattn_fused_stamps": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sd_op_gemm_fp8": (_i, [_vp, _vp, _ll, _vp, _vp, _f, _vp, _vp, _ll, _vp, _ll, _i, _i, _i, _i, _i, _f]),
    "sd_op_conv3x3_fp8": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    "sd_op_groupnorm_fp8": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _i, _i, _f]),
    "sd_op_layernorm_fp8": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f]),
    "sd_op_quantize_fp8": (_i, [_vp, _vp, _vp, _ll, _i, _i, _f]),
}


def load() -> C.CDLL:
    """Load libsdhip.so (RTLD_GLOBAL not needed); raises SdHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SdHipError(
            f"{LIB_PATH} is missing: build it with `python -m sonicdiffusionbayeslab_amd.build` "
            "(or __graft_entry__.build()).  There is no CPU/PyTorch fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def load_ablate() -> C.CDLL:
    """libsdhip_ablate.so: a SECOND, independent instance of the library built with -DSD_ABLATE (its own globals).  Never
    used by the product path; raises if it is not built."""
    global _ablate
    if _ablate is not None:
        return _ablate
    if not os.path.exists(ABLATE_LIB_PATH):
        raise SdHipError(f"{ABLATE_LIB_PATH} is missing: build it with `python -m sonicdiffusionbayeslab_amd.build`")
    lib = C.CDLL(ABLATE_LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _ablate = lib
    return lib


def check(rc: int, what: str = "libsdhip call", lib: Optional[C.CDLL] = None) -> None:
    if rc != 0:
        msg = (lib or load()).sd_last_error()
        raise SdHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t) -> Optional[int]:
    """Device/host pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
