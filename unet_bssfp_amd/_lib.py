"""ctypes binding of the C-ABI library ``libmi355_unet.so`` (include/mi355_unet.h).

The product path has NO fallback: if the library is missing or a call fails this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi355_unet.so")   # the one shipped build; no environment override (tools/diaglib.py swaps it for A/B runs)

DT_F32 = 0
DT_BF16 = 1
DT_FP8 = 3

_i32, _i64, _f32, _u64, _vp = C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_void_p


class WpackDesc(C.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("cout", _i32), ("cin", _i32), ("coutp", _i32), ("cinp", _i32),
                ("ks", _i32), ("s_co", _i64), ("s_ci", _i64), ("s_k", _i64 * 3),
                ("tbase", _i32 * 3), ("tstep", _i32 * 3), ("dtype", _i32), ("s2d_mode", _i32), ("s2d_cp", _i32),
                ("q_amax", _vp)]


class ConvDesc(C.Structure):
    _fields_ = [("x0", _vp), ("c0", _i32), ("ld0", _i32), ("x1", _vp), ("c1", _i32), ("ld1", _i32),
                ("n", _i32), ("di", _i32), ("hi", _i32), ("wi", _i32),
                ("do_", _i32), ("ho", _i32), ("wo", _i32), ("ks", _i32), ("stride", _i32),
                ("pad", _i32 * 3), ("wp", _vp), ("coutp", _i32), ("bias", _vp),
                ("y", _vp), ("ldy", _i32), ("cstore", _i32), ("dy", _i32), ("hy", _i32), ("wy", _i32),
                ("os", _i32), ("ooff", _i32 * 3), ("stats_part", _vp), ("dtype", _i32),
                ("workspace", _vp), ("workspace_bytes", _i64), ("cls_cout", _i32), ("nbias", _i32),
                ("q_amax_x", _vp), ("q_amax_w", _vp), ("addend", _vp), ("ld_add", _i32), ("y_f32", _i32), ("add_n", _i32),
                ("d2s", _i32), ("delta", _vp), ("add_bf16", _i32)]


class WgradDesc(C.Structure):
    _fields_ = [("x0", _vp), ("c0", _i32), ("ld0", _i32), ("x1", _vp), ("c1", _i32), ("ld1", _i32),
                ("n", _i32), ("di", _i32), ("hi", _i32), ("wi", _i32),
                ("g", _vp), ("cg", _i32), ("ldg", _i32),
                ("do_", _i32), ("ho", _i32), ("wo", _i32), ("gd", _i32), ("gh", _i32), ("gw", _i32),
                ("gs", _i32), ("goff", _i32 * 3), ("ks", _i32), ("stride", _i32), ("pad", _i32 * 3),
                ("workspace", _vp), ("workspace_bytes", _i64),
                ("dw", _vp), ("cout", _i32), ("cin", _i32), ("s_co", _i64), ("s_ci", _i64), ("s_k", _i64 * 3),
                ("tbase", _i32 * 3), ("tstep", _i32 * 3), ("accumulate", _i32), ("dtype", _i32), ("s2d_cp", _i32),
                ("g_cls_cout", _i32), ("xn", _i32)]


class WreduceJob(C.Structure):
    _fields_ = [("opaque", _i64 * 20)]


class NormActDesc(C.Structure):
    _fields_ = [("z", _vp), ("ldz", _i32), ("a", _vp), ("lda", _i32), ("c", _i32),
                ("rows_per_group", _i64), ("groups", _i32),
                ("mean", _vp), ("rstd", _vp), ("gamma", _vp), ("beta", _vp),
                ("slope", _f32), ("drop_p", _f32), ("seed", _u64), ("dtype", _i32),
                ("da", _vp), ("ldda", _i32), ("dz", _vp), ("lddz", _i32),
                ("part", _vp), ("blocks_per_group", _i32), ("sums", _vp), ("batch_stats", _i32),
                ("s2d_a", _i32), ("s2d_da", _i32), ("sd", _i32), ("sh", _i32), ("sw", _i32), ("seed_ptr", _vp),
                ("n_affine", _i32), ("q8", _vp), ("ld8", _i32), ("q_use", _vp), ("q_next", _vp),
                ("gz", _vp), ("ldgz", _i32), ("gw", _vp), ("gw_ld", _i32), ("gk", _i32),
                ("fy", _vp), ("ldfy", _i32), ("fcp", _i32), ("fbias", _vp), ("skip_a", _i32),
                ("pool_idx", _vp), ("pool_dy", _vp), ("ldpdy", _i32),
                ("pool_y", _vp), ("ldpy", _i32), ("pool_widx", _vp)]


class NormSmallDesc(C.Structure):
    _fields_ = [("base", NormActDesc), ("eps", _f32), ("momentum", _f32), ("mean_out", _vp), ("rstd_out", _vp),
                ("running_mean", _vp), ("running_var", _vp), ("batches_tracked", _vp), ("n_real", _i32),
                ("dgamma", _vp), ("dbeta", _vp), ("accumulate", _i32)]


_SIGNATURES = {
    "mi355_version": (C.c_int, []),
    "mi355_last_error": (C.c_char_p, []),
    "mi355_pack_ncdhw": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mi355_unpack_ncdhw": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _i32, _i32, _i32, _vp]),
    "mi355_pack_ncdhw_s2d": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mi355_unpack_ncdhw_s2d": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mi355_pack2_ncdhw": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _vp]),
    "mi355_pack2_ncdhw_s2d": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mi355_upcat_compose": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mi355_upcat_chain": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "mi355_border_sums_workspace": (_i64, [_i32, _i32, _i32]),
    "mi355_border_sums": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "mi355_s2d_repack": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mi355_seam_grad": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mi355_weight_pack": (C.c_int, [C.POINTER(WpackDesc), _vp]),
    "mi355_weight_pack_multi": (C.c_int, [C.POINTER(WpackDesc), _i32, _vp]),
    "mi355_conv_fwd": (C.c_int, [C.POINTER(ConvDesc), _vp]),
    "mi355_conv_workspace_bytes": (_i64, [C.POINTER(ConvDesc)]),
    "mi355_conv_plan_id": (C.c_int, [C.POINTER(ConvDesc)]),
    "mi355_conv_num_tiles": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(_i32), C.POINTER(_i32)]),
    "mi355_conv_wgrad_workspace": (_i64, [C.POINTER(WgradDesc)]),
    "mi355_conv_wgrad_plan_kind": (C.c_int, [C.POINTER(WgradDesc)]),
    "mi355_conv_wgrad": (C.c_int, [C.POINTER(WgradDesc), _vp]),
    "mi355_conv_wgrad_partial": (C.c_int, [C.POINTER(WgradDesc), C.POINTER(WreduceJob), _vp]),
    "mi355_wgrad_reduce_multi": (C.c_int, [C.POINTER(WreduceJob), _i32, _vp]),
    "mi355_channel_stats": (C.c_int, [_vp, _i32, _i32, _i64, _i32, _vp, _i32, _i32, _vp]),
    "mi355_channel_stats_blocks": (_i32, [_i64]),
    "mi355_norm_finalize": (C.c_int, [_vp, _i32, _i32, _i32, _i64, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _f32, _vp, _vp]),
    "mi355_normact_fwd": (C.c_int, [C.POINTER(NormActDesc), _vp]),
    "mi355_normact_bwd_reduce": (C.c_int, [C.POINTER(NormActDesc), _vp]),
    "mi355_normact_bwd_finalize": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mi355_normact_bwd_finalize_into": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp]),
    "mi355_normact_bwd_apply": (C.c_int, [C.POINTER(NormActDesc), _vp]),
    "mi355_normact_small_fwd": (C.c_int, [C.POINTER(NormSmallDesc), _vp]),
    "mi355_normact_small_bwd": (C.c_int, [C.POINTER(NormSmallDesc), _vp]),
    "mi355_colsum_finalize": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "mi355_colsum_finalize_into": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _vp]),
    "mi355_colsum_finalize_from": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "mi355_maxpool2_fwd": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mi355_maxpool2_fwd_idx": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mi355_maxpool2_bwd": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mi355_maxpool2_bwd_add": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "mi355_l1_blocks": (_i32, [_i64]),
    "mi355_l1_fwd": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "mi355_l1_bwd": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "mi355_l1_partials": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "mi355_gan_gen_loss_fwd": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _f32, _f32, _vp, _vp]),
    "mi355_gan_gen_loss_bwd": (C.c_int, [_vp, _i32, _vp, _f32, _f32, _vp, _vp, _vp]),
    "mi355_gan_discr_loss_fwd": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp]),
    "mi355_gan_discr_loss_bwd": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "mi355_adamw_multi": (C.c_int, [_vp, _vp, _i32, _f32, _f32, _f32, _f32, _f32, _vp, _i64, _vp]),
    "mi355_dti_scalar_maps": (C.c_int, [_vp, _i32, _i64, _i64, _i64, C.c_double, C.c_double,
                                        _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mi355_patch_gather": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mi355_patch_aggregate": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mi355_patch_average_finalize": (C.c_int, [_vp, _vp, _i32, _i64, _vp]),
    "mi355_err_blocks": (_i32, [_i64]),
    "mi355_err_sums": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "mi355_ssim3d_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "mi355_ssim3d": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _f32, _f32, _vp, _i64, _vp, _vp]),
    "mi355_aug_bias_field": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "mi355_aug_gamma": (C.c_int, [_vp, _vp, _i64, _f32, _vp]),
    "mi355_aug_noise": (C.c_int, [_vp, _vp, _i64, _f32, _f32, C.c_uint64, _vp]),
    "mi355_mfma_selftest": (C.c_int, [_vp, _vp, _vp]),
    "mi355_amax_f32": (C.c_int, [_vp, _i64, _vp, _vp]),
    "mi355_amax_act": (C.c_int, [_vp, _i32, _i32, _i64, _i32, _vp, _vp]),
    "mi355_cast_fp8": (C.c_int, [_vp, _i32, _i32, _i64, _i32, _vp, _vp, _i32, _vp]),
    "mi355_cast_fp8_delayed": (C.c_int, [_vp, _i32, _i32, _i64, _i32, _vp, _vp, _vp, _i32, _vp]),
    "mi355_fp8_scale_roll": (C.c_int, [_vp, _i32, _vp, _vp]),
    "mi355_fp8_selftest": (C.c_int, [_vp, _vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lock = threading.Lock()
_lib = None


class Mi355Error(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library object.  Raises if the .so is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise Mi355Error(
                    f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().mi355_last_error().decode("utf-8", "replace")
        raise Mi355Error(f"{what or 'mi355 call'} failed ({rc}): {msg}")
