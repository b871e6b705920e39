"""ctypes binding of libtmdiff_hip.so (include/tmdiff_hip.h).

The library is the product: if it is missing or fails to load, importing this module
raises -- there is no CPU or PyTorch fallback behind it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TMDIFF_HIP_LIB") or os.path.join(_HERE, "libtmdiff_hip.so")   # override: diagnostic builds

c_float_p = C.POINTER(C.c_float)
vp = C.c_void_p


class Conv3dDesc(C.Structure):
    """struct tmdiff_conv3d_desc"""
    _fields_ = [
        ("B", C.c_int32), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("groups", C.c_int32), ("ksize", C.c_int32), ("nseg", C.c_int32),
        ("seg_c", C.c_int32 * 3),
        ("seg_x", vp * 3),
        ("w_packed", vp), ("bias", vp), ("bias_scale", C.c_float),
        ("in_shift", vp), ("in_scale", vp), ("in_shift_stride", C.c_int32), ("in_scale_stride", C.c_int32),
        ("in_mask", vp), ("in_act", C.c_int32),
        ("residual", vp), ("out_scale", C.c_float),
        ("y", vp),
        ("y2", vp), ("y2_shift", vp), ("y2_scale", vp), ("y2_shift_stride", C.c_int32), ("y2_scale_stride", C.c_int32),
        ("y2_act", C.c_int32), ("y2_bf16", C.c_int32), ("x_bf16", C.c_int32),
        ("splitk_ws", vp), ("splitk_ws_bytes", C.c_int64),
        ("drop_seed", C.c_uint64), ("drop_p", C.c_float), ("drop_seed_dev", vp), ("rc_x", vp), ("rc_w", vp), ("rc_cin", C.c_int32), ("y_ll", vp), ("y_hi", vp * 3), ("xp_out", vp), ("xp_shift", vp), ("xp_shift_stride", C.c_int32), ("xp_act", C.c_int32),
        ("y2_s2d", C.c_int32),
    ]


# name -> (restype, argtypes); mirrors include/tmdiff_hip.h one to one
SIGNATURES = {
    "tmdiff_version": (C.c_int, []),
    "tmdiff_last_error_string": (C.c_char_p, []),
    "tmdiff_conv3d_pack_weights": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]),
    "tmdiff_conv3d_pack_weights_multi_chunks": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "tmdiff_conv3d_pack_weights_multi": (C.c_int, [vp, vp, vp, C.c_int32, vp]),
    "tmdiff_conv3d_fwd": (C.c_int, [C.POINTER(Conv3dDesc), vp]),
    "tmdiff_conv3d_fwd_splitk_workspace_bytes": (C.c_size_t, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_fwd_xp_supported": (C.c_int, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_fwd_staged_supported": (C.c_int, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_fwd_staged_workspace_bytes": (C.c_size_t, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_fwd_staged": (C.c_int, [C.POINTER(Conv3dDesc), vp, vp]),
    "tmdiff_conv3d_wino_supported": (C.c_int, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_wino_blocks": (C.c_int64, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_wino_workspace_bytes": (C.c_size_t, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_wino_planes": (C.c_int32, [C.c_int32]),
    "tmdiff_conv3d_wino_packed_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "tmdiff_conv3d_wino_pack_weights": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]),
    "tmdiff_conv3d_wino_pack_weights_multi_chunk": (C.c_int32, []),
    "tmdiff_conv3d_wino_pack_weights_multi": (C.c_int, [vp, vp, vp, C.c_int32, vp]),
    "tmdiff_conv3d_wf_supported": (C.c_int, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_wf_blocks": (C.c_int64, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_wf_workspace_bytes": (C.c_size_t, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_wf_splitk_workspace_bytes": (C.c_size_t, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_wf_fwd": (C.c_int, [C.POINTER(Conv3dDesc), vp, vp]),
    "tmdiff_conv3d_wino_fwd": (C.c_int, [C.POINTER(Conv3dDesc), vp, vp]),
    "tmdiff_conv3d_wino_fwd_planes": (C.c_int, [C.POINTER(Conv3dDesc), vp, C.c_int32, vp, C.c_int32, vp]),
    "tmdiff_conv3d_wino_fwd_xp": (C.c_int, [C.POINTER(Conv3dDesc), vp, C.c_int32, vp, vp]),
    "tmdiff_conv3d_wino_fwd_stage": (C.c_int, [C.POINTER(Conv3dDesc), vp, C.c_int32, vp]),
    "tmdiff_conv3d_prologue_fwd": (C.c_int, [C.POINTER(Conv3dDesc), vp, vp]),
    "tmdiff_conv3d_ll_supported": (C.c_int, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_ll_splitk_workspace_bytes": (C.c_size_t, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_ll_packed_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "tmdiff_conv3d_ll_pack_weights": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_float, vp]),
    "tmdiff_conv3d_ll_fwd": (C.c_int, [C.POINTER(Conv3dDesc), C.c_float, vp]),
    "tmdiff_conv3d_packed_bf16_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "tmdiff_conv3d_pack_weights_bf16": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]),
    "tmdiff_conv3d_bf16_workspace_bytes": (C.c_size_t, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_fwd_bf16": (C.c_int, [C.POINTER(Conv3dDesc), vp, vp]),
    "tmdiff_conv3d_wgrad_workspace_bytes": (C.c_size_t, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_wgrad": (C.c_int, [C.POINTER(Conv3dDesc), vp, vp, vp, vp]),
    "tmdiff_conv3d_wgrad_bias": (C.c_int, [C.POINTER(Conv3dDesc), vp, vp, vp, vp, vp]),
    "tmdiff_conv3d_wf_plan": (C.c_int32, [C.c_int32] * 8 + [C.POINTER(C.c_int64)]),
    "tmdiff_conv3d_wfll_supported": (C.c_int, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_wfll_packed_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "tmdiff_conv3d_wfll_pack_weights": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_float, vp]),
    "tmdiff_conv3d_wfll_splitk_workspace_bytes": (C.c_size_t, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_wfll_fwd": (C.c_int, [C.POINTER(Conv3dDesc), C.c_float, vp]),
    "tmdiff_conv3d_wgrad_wino_supported": (C.c_int, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_wgrad_wino_workspace_bytes": (C.c_size_t, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_wgrad_wino": (C.c_int, [C.POINTER(Conv3dDesc), vp, vp, vp, vp]),
    "tmdiff_conv3d_wgrad_wino_bias": (C.c_int, [C.POINTER(Conv3dDesc), vp, vp, vp, vp, vp]),
    "tmdiff_channel_sum": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_float, vp]),
    "tmdiff_conv3d_prologue_bwd": (C.c_int, [C.POINTER(Conv3dDesc), vp, vp * 3, C.c_int32 * 3, vp, vp, vp]),
    "tmdiff_conv3d_prologue_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(Conv3dDesc)]),
    "tmdiff_conv3d_prologue_bwd_ws": (C.c_int, [C.POINTER(Conv3dDesc), vp, vp * 3, C.c_int32 * 3, vp, vp, vp, vp]),
    "tmdiff_conv3d_prologue_bwd_add": (C.c_int, [C.POINTER(Conv3dDesc), vp, vp * 3, vp * 3, vp, vp, vp, vp]),
    "tmdiff_stem_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]),
    "tmdiff_stem_bwd_input": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        vp]),
    "tmdiff_head_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int64, vp]),
    "tmdiff_linear_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]),
    "tmdiff_stem_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, vp]),
    "tmdiff_head_fwd": (C.c_int, [vp, vp, vp, C.c_int32, vp, C.c_int32, C.c_int32, C.c_int64, vp]),
    "tmdiff_haar_dwt2d": (C.c_int, [vp, vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float, vp]),
    "tmdiff_haar_idwt2d": (C.c_int, [vp * 2, C.c_int32, vp, vp, vp, C.c_int64, C.c_int64, vp * 2, C.c_int64,
                                     C.c_int32, C.c_int32, C.c_float, vp]),
    "tmdiff_linear_fwd": (C.c_int, [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]),
    "tmdiff_gamma_embedding": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int32, vp]),
    "tmdiff_ddpm_step": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_int32, vp]),
    "tmdiff_axpby": (C.c_int, [vp * 4, C.c_float * 4, C.c_int32, vp, C.c_int64, vp]),
    "tmdiff_multi_axpby_chunk": (C.c_int32, []),
    "tmdiff_multi_axpby": (C.c_int, [vp, vp, vp, C.c_int32, C.c_float, C.c_float, vp]),
    "tmdiff_multi_adamw": (C.c_int, [vp, vp, vp, C.c_int32, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
    "tmdiff_x0_from_model": (C.c_int, [vp, vp, vp, C.c_int64, C.c_float, C.c_float, C.c_int32, vp]),
    "tmdiff_abs_quantile_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int64]),
    "tmdiff_abs_quantile_clamp": (C.c_int, [vp, C.c_int32, C.c_int64, C.c_float, C.c_float, vp, vp]),
    "tmdiff_add": (C.c_int, [vp, vp, vp, C.c_int64, C.c_float, vp]),
    "tmdiff_attn_fwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int64 * 3, C.c_int64 * 3, C.c_int64 * 3, C.c_int64 * 3, C.c_float, vp]),
    "tmdiff_gemm_nt": (C.c_int, [vp, vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, vp]),
    "tmdiff_group_norm": (C.c_int, [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_float, vp]),
    "tmdiff_layer_norm": (C.c_int, [vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_float, vp]),
    "tmdiff_geglu": (C.c_int, [vp, vp, C.c_int64, C.c_int32, C.c_int32, vp]),
    "tmdiff_q_sample": (C.c_int, [vp, vp, vp, vp, C.c_int32, C.c_int64, vp]),
    # per-operator names (thin fronts of the entry points above)
    "tmdiff_conv3d_k3_fwd": (C.c_int, [C.POINTER(Conv3dDesc), vp]),
    "tmdiff_conv3d_k3_dgrad": (C.c_int, [C.POINTER(Conv3dDesc), vp]),
    "tmdiff_conv3d_k3_wgrad": (C.c_int, [C.POINTER(Conv3dDesc), vp, vp, vp, vp]),
    "tmdiff_conv3d_k1_fwd": (C.c_int, [C.POINTER(Conv3dDesc), vp]),
    "tmdiff_conv3d_k1_dgrad": (C.c_int, [C.POINTER(Conv3dDesc), vp]),
    "tmdiff_conv3d_k1_wgrad": (C.c_int, [C.POINTER(Conv3dDesc), vp, vp, vp, vp]),
    "tmdiff_haar_dwt2d_fwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float, vp]),
    "tmdiff_haar_dwt2d_bwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float, vp]),
    "tmdiff_haar_idwt2d_fwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, C.c_float, vp]),
    "tmdiff_haar_idwt2d_bwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, C.c_float, vp]),
    "tmdiff_dpm_axpby2": (C.c_int, [vp, C.c_float, vp, C.c_float, vp, C.c_int64, vp]),
    "tmdiff_dpm_axpby3": (C.c_int, [vp, C.c_float, vp, C.c_float, vp, C.c_float, vp, C.c_int64, vp]),
    "tmdiff_dpm_axpby4": (C.c_int, [vp, C.c_float, vp, C.c_float, vp, C.c_float, vp, C.c_float, vp, C.c_int64, vp]),
}


class PlanePrologue(C.Structure):
    """struct tmdiff_plane_prologue"""
    _fields_ = [("shift", vp), ("scale", vp), ("shift_stride", C.c_int32), ("scale_stride", C.c_int32),
                ("C", C.c_int32), ("n_per_channel", C.c_int32), ("act", C.c_int32)]


SIGNATURES.update({
    "tmdiff_haar_dwt2d_pro": (C.c_int, [vp, vp, vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                        C.POINTER(PlanePrologue), vp]),
    "tmdiff_haar_idwt2d_pro": (C.c_int, [vp * 2, C.c_int32, vp, vp, vp, C.c_int64, C.c_int64, vp * 2, C.c_int64,
                                         C.c_int32, C.c_int32, C.c_float, C.POINTER(PlanePrologue), vp]),
    "tmdiff_haar_dwt2d_pack_bf16": (C.c_int, [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_float, C.c_float, C.POINTER(PlanePrologue), vp]),
    "tmdiff_haar_idwt2d_pack_bf16": (C.c_int, [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                               C.c_float, C.POINTER(PlanePrologue), vp]),
    "tmdiff_stem_fwd_pack_bf16": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int32, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_int32, vp]),
    "tmdiff_stem_fwd_scaled": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int32, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, vp]),
})


class TmdiffError(RuntimeError):
    """Non-zero status from the C ABI (the reference surfaces errors as ordinary exceptions)."""


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "tmdiff_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == header/library mismatch
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()
ABI_VERSION = lib.tmdiff_version()


def check(status, what=""):
    if status != 0:
        msg = lib.tmdiff_last_error_string().decode("utf-8", "replace")
        raise TmdiffError(f"{what or 'tmdiff'} failed with status {status}: {msg}")
