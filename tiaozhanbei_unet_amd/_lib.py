"""ctypes binding of libunet_hip.so (C-ABI declared in include/unet_hip.h).

The library is the ONLY compute path of this package: there is no CPU or eager
PyTorch fallback.  If the shared object is missing or a call fails, a RuntimeError
is raised -- loudly -- instead of silently computing something else.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UNET_HIP_LIB") or os.path.join(_HERE, "libunet_hip.so")   # (override: diagnostic builds)
CSRC = os.path.join(_HERE, "csrc")

UNET_F32, UNET_BF16 = 0, 1
PACK_CONV_FWD, PACK_CONV_DGRAD, PACK_CONVT_FWD, PACK_CONVT_DGRAD = 0, 1, 2, 3
(K_CONV_FWD, K_CONV_DGRAD, K_CONV_WGRAD, K_CONVT_FWD, K_CONVT_DGRAD, K_CONVT_WGRAD, K_BN, K_POOL,
 K_HEAD, K_LOSS, K_PACK, K_OTHER, K_COUNT) = range(13)
KCLASS_NAMES = ["conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad", "convt_fwd", "convt_dgrad",
                "convt_wgrad", "bn", "pool_upsample", "head", "loss", "pack_layout", "other"]


class View(C.Structure):
    """struct unet_view"""
    _fields_ = [("ptr", C.c_void_p), ("c", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("off_y", C.c_int32), ("off_x", C.c_int32)]


View2 = View * 2
_i, _l, _f, _p, _z, _d = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t, C.c_double

# name -> (restype, argtypes): every symbol include/unet_hip.h declares
SIGNATURES = {
    "unet_abi_version": (_i, []),
    "unet_last_error": (C.c_char_p, []),
    "unet_tuning_reload": (_i, []),
    "unet_set_reserved_cus": (_i, [_i]),
    "unet_get_cu_budget": (_i, []),
    "unet_debug_spin": (_i, [_i, _i, _i, _p]),
    "unet_prof_enable": (_i, [_i]),
    "unet_prof_collect": (_i, [_p, _p, _p]),
    "unet_prof_kernel_stats": (_i, [_i, _p, _p, _p, _p]),
    "unet_prof_kernel_bytes": (_i, [_i, _p]),
    "unet_nchw_to_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "unet_nhwc_to_nchw": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "unet_pack_weight": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "unet_pack_weights_batched": (_i, [_p, _i, _i, _p]),
    "unet_conv3x3": (_i, [_i, _i, _i, _i, C.POINTER(View), _p, _i, C.POINTER(View), _i, _i, _i, _p]),
    "unet_pack_conv_weight_folded": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "unet_conv3x3_bias_relu": (_i, [_i, _i, _i, _i, C.POINTER(View), _p, _i, _p, _p, _i, _p]),
    "unet_conv3x3_stats_max_parts": (_z, [_i, _i, _i]),
    "unet_conv3x3_stats": (_i, [_i, _i, _i, _i, C.POINTER(View), _p, _i, _p, _p, _p, _p]),
    "unet_conv3x3_first_supported": (_i, [_i, _i, _i, _i]),
    "unet_conv3x3_first_stats": (_i, [_i, _i, _i, _p, _i, _p, _p, _p, _p, _p]),
    "unet_conv3x3_first_wgrad_workspace": (_z, [_i, _i, _i]),
    "unet_conv3x3_first_wgrad": (_i, [_i, _i, _i, _p, _i, _p, _p, _p, _z, _p]),
    "unet_conv3x3_first_wgrad_bn": (_i, [_i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _z, _p]),
    "unet_bn_finalize_partials": (_i, [_p, _i, _l, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p]),
    "unet_convt2x2_dgrad_bnrelu_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "unet_convt2x2_dgrad_bnrelu_max_parts": (_z, []),
    "unet_convt2x2_dgrad_bnrelu": (_i, [_i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "unet_conv3x3_dgrad_bnrelu_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "unet_conv3x3_dgrad_bnrelu": (_i, [_i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "unet_conv3x3_wgrad_workspace": (_z, [_i, _i, _i, _i, _i]),
    "unet_conv3x3_wgrad": (_i, [_i, _i, _i, _i, C.POINTER(View), _p, _i, _p, _i, _p, _z, _p]),
    "unet_convt2x2_fwd": (_i, [_i, _i, _i, _i, _p, _i, _p, _p, _p, _i, _p]),
    "unet_convt2x2_dgrad": (_i, [_i, _i, _i, _i, _p, _i, _p, _p, _i, _p]),
    "unet_convt2x2_wgrad_workspace": (_z, [_i, _i, _i, _i, _i]),
    "unet_convt2x2_wgrad": (_i, [_i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _z, _p]),
    "unet_bn_workspace": (_z, [_l, _i]),
    "unet_bn_train_stats": (_i, [_i, _p, _l, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _z, _p]),
    "unet_bn_eval_coeffs": (_i, [_i, _p, _p, _p, _p, _f, _p, _p, _p]),
    "unet_bn_eval_coeffs4": (_i, [_i, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p]),
    "unet_bn_relu_bwd_frozen": (_i, [_i, _p, _p, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "unet_bn_relu_apply": (_i, [_i, _p, _l, _i, _p, _p, _p, _p]),
    "unet_bn_relu_bwd": (_i, [_i, _p, _p, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "unet_bn_bwd_premasked": (_i, [_i, _p, _p, _l, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _z, _p]),
    "unet_head_bnrelu_fwd": (_i, [_i, _p, _i, _i, _i, _i, _p, _p, _p, _p, _i, _i, _p, _p]),
    "unet_head_bnrelu_max_parts": (_z, []),
    "unet_head_bnrelu_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _z, _p]),
    "unet_maxpool2_fwd": (_i, [_i, _p, _i, _i, _i, _i, _p, _p]),
    "unet_maxpool2_bwd": (_i, [_i, _p, _p, _i, _i, _i, _i, _p, _i, _p]),
    "unet_bn_relu_pool_supported": (_i, [_i, _i]),
    "unet_bn_relu_pool_fwd": (_i, [_i, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "unet_bn_relu_pool_max_parts": (_z, []),
    "unet_bn_relu_pool_bwd": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "unet_upsample_bilinear2x_fwd": (_i, [_i, _p, _i, _i, _i, _i, _p, _p]),
    "unet_upsample_bilinear2x_bwd": (_i, [_i, _p, _i, _i, _i, _i, _p, _p]),
    "unet_head_fwd": (_i, [_i, _p, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p]),
    "unet_head_bwd_workspace": (_z, [_i, _i, _i, _i, _i]),
    "unet_head_bwd": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _z, _p]),
    "unet_loss_workspace": (_z, [_l]),
    "unet_loss_mse_focal": (_i, [_p, _p, _l, _p, _p, _l, _f, _f, _p, _p, _p, _p, _z, _p]),
    "unet_ssim_workspace": (_z, [_i, _i, _i]),
    "unet_ssim_loss": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _z, _p]),
    "unet_ssim_loss_per_image": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _z, _p]),
    "unet_seg_loss_workspace": (_z, [_i, _i, _l]),
    "unet_seg_loss": (_i, [_p, _p, _i, _i, _l, _p, _l, _i, _f, _f, _f, _f, _f, _p, _p, _p, _z, _p]),
    "unet_seg_confusion": (_i, [_p, _p, _i, _i, _l, _l, _p, _p, _p]),
    "unet_threshold_confusion": (_i, [_p, _p, _p, _l, _l, _p, _i, _p, _p]),
    "unet_channel_scale": (_i, [_i, _p, _p, _i, _l, _i, _p, _p]),
    "unet_anomaly_score_workspace": (_z, [_i, _l]),
    "unet_anomaly_score": (_i, [_p, _p, _i, _i, _l, _i, _p, _p, _p, _z, _p]),
    "unet_preprocess_u8": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "unet_resize_bilinear_ksize": (_i, [_i, _i]),
    "unet_resize_bilinear_coeffs": (_i, [_i, _i, _p, _p]),
    "unet_resize_bilinear_u8": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p, _p, _i, _p, _p, _p]),
    "unet_resize_nearest_index": (_i, [_i, _i, _p]),
    "unet_resize_nearest_u8": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "unet_flip_rotate_u8": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "unet_color_jitter_workspace": (_z, [_i]),
    "unet_color_jitter_normalize_u8": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p, _z, _p]),
    "unet_adam_chunk_elems": (_i, []),
    "unet_adam_multi": (_i, [_p, _p, _i, _f, _d, _d, _f, _f, _f, _i, _i, _p]),
    "unet_adam_step": (_i, [_p, _p, _p, _p, _l, _f, _d, _d, _f, _f, _f, _i, _p]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile libunet_hip.so for gfx950 with hipcc (csrc/Makefile).  Cross-compiles without a GPU."""
    if force or not os.path.exists(LIB_PATH) or _stale():
        subprocess.run(["make", "-C", CSRC, "-j8"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def _stale() -> bool:
    if not os.path.isdir(CSRC):
        return False
    t = os.path.getmtime(LIB_PATH)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "unet_hip.h"))
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in srcs)


def lib():
    """The loaded library; raises if it cannot be loaded (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libunet_hip.so not found at {LIB_PATH}: the HIP extension is the only compute path of "
                "tiaozhanbei_unet_amd (no CPU/eager fallback). Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C tiaozhanbei_unet_amd/csrc`.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the .so does not export it
            fn.restype, fn.argtypes = res, args
        if handle.unet_abi_version() != 1:
            raise RuntimeError("libunet_hip.so: ABI version mismatch")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().unet_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (status {rc}): {msg}")
