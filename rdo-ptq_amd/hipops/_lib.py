"""ctypes binding of librdoptq_hip.so (the C ABI declared in include/rdo_ptq_hip.h).

The library is the only compute back end of this package: if it cannot be loaded, or a call fails, a RuntimeError is
raised -- there is no CPU or eager-PyTorch fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "librdoptq_hip.so")

EPI_NONE, EPI_LRELU, EPI_LRELU_BWD, EPI_GDN, EPI_IGDN, EPI_RELU, EPI_RELU_BWD, EPI_GELU, EPI_GELU_BWD = range(9)
LOG_SLOTS = 32            # RDO_LOG_SLOTS of include/rdo_ptq_hip.h


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "H", "W", "Cin", "Ho", "Wo", "Cout", "KH", "KW", "stride", "pad", "epilogue",
                                         "square_input", "add_residual")]


class AdaDesc(C.Structure):
    _fields_ = [("numel", C.c_int64), ("rows", C.c_int32), ("n_levels", C.c_int32), ("reparam", C.c_int32),
                ("reparam_bound", C.c_float), ("reparam_pedestal", C.c_float), ("KH", C.c_int32), ("KW", C.c_int32),
                ("Cin", C.c_int32)]


class AdaStepItem(C.Structure):
    _fields_ = [("d", AdaDesc), ("w", C.c_void_p), ("delta", C.c_void_p), ("zp", C.c_void_p), ("slabs", C.c_void_p),
                ("nsplit", C.c_int32), ("alpha", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("wq", C.c_void_p),
                ("wd", C.c_void_p), ("wq_planes", C.c_void_p), ("wd_planes", C.c_void_p), ("dalpha", C.c_void_p),
                ("wq_plane_scale", C.c_float), ("wd_plane_scale", C.c_float),
                ("lin_fwd_planes", C.c_void_p), ("lin_bwd_planes", C.c_void_p), ("lin_plane_scale", C.c_float)]


class GatherDesc(C.Structure):
    _fields_ = [("cache_q", C.c_void_p), ("cache_fp", C.c_void_p), ("idx_table", C.c_void_p), ("n_iters", C.c_int32), ("B", C.c_int32),
                ("batch_offset", C.c_int32), ("per_image", C.c_int64), ("C", C.c_int32), ("prob", C.c_float), ("seed", C.c_uint32),
                ("out", C.c_void_p), ("out_planes", C.c_void_p), ("out_scale", C.c_float), ("overflow_flag", C.c_void_p)]


class AttnDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "H", "W", "C", "heads", "window", "shift")] + [("scale", C.c_float)]


class SchedRow(C.Structure):
    _fields_ = [("b", C.c_float), ("round_on", C.c_float), ("step_size", C.c_float), ("bc2_sqrt", C.c_float)]


P = C.c_void_p
_SIGS = {
    "rdo_version": (C.c_char_p, []),
    "rdo_last_error": (C.c_char_p, []),
    "rdo_set_tuning": (C.c_int, [C.c_char_p, C.c_int32]),
    "rdo_get_tuning": (C.c_int, [C.c_char_p]),
    "rdo_conv2d_fwd": (C.c_int, [C.POINTER(ConvDesc), P, P, P, P, P, P, P, P, C.c_int64, P, P]),
    "rdo_conv2d_fwd_uses_bf16x6": (C.c_int, [C.POINTER(ConvDesc)]),
    "rdo_split_bf16x3": (C.c_int, [P, C.c_int64, P, P]),
    "rdo_split_bf16x3_conv": (C.c_int, [P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, P]),
    "rdo_conv2d_fwd_bf16x6": (C.c_int, [C.POINTER(ConvDesc), P, P, P, P, P, P, P, P, C.c_int64, P]),
    "rdo_conv2d_fwd_bf16x6_ksplit": (C.c_int, [C.POINTER(ConvDesc)]),
    "rdo_conv2d_fwd_workspace": (C.c_int64, [C.POINTER(ConvDesc)]),
    "rdo_conv2d_wgrad_nsplit": (C.c_int, [C.POINTER(ConvDesc)]),
    "rdo_conv2d_wgrad": (C.c_int, [C.POINTER(ConvDesc), P, P, P, C.c_int, P]),
    "rdo_conv2d_wgrad_uses_bf16x6": (C.c_int, [C.POINTER(ConvDesc)]),
    "rdo_reduce_slabs": (C.c_int, [P, C.c_int, C.c_int64, P, P]),
    "rdo_adaround_init_alpha": (C.c_int, [C.POINTER(AdaDesc), P, P, P, P]),
    "rdo_adaround_fwd": (C.c_int, [C.POINTER(AdaDesc), P, P, P, P, C.c_int, P, P, P]),
    "rdo_adaround_step": (C.c_int, [C.POINTER(AdaDesc), P, P, P, P, C.c_int, C.c_float, C.c_float, P, P, P, P, P, P, P, P, P, P, C.c_float, C.c_float, P]),
    "rdo_adaround_step_batch": (C.c_int, [C.POINTER(AdaStepItem), C.c_int32, C.c_int32, C.c_float, C.c_float, P, P, P, P, P, P]),
    "rdo_adaround_step_batch_gather": (C.c_int, [C.POINTER(AdaStepItem), C.c_int32, C.c_int32, C.c_float, C.c_float, P, P, P, P, C.POINTER(GatherDesc), P]),
    "rdo_iter_bind_publish": (C.c_int, [P]),
    "rdo_unit1x1_supported": (C.c_int, [C.c_int64, C.c_int32, C.c_int32]),
    "rdo_unit1x1_nslab": (C.c_int, [C.c_int64, C.c_int32]),
    "rdo_unit1x1_form": (C.c_int, [C.c_int32]),
    "rdo_unit1x1": (C.c_int, [P, C.c_int64, C.c_int32, C.c_int32, P, P, P, P, P, C.c_int32, C.c_float, C.c_int32, P, C.c_int32, P, P]),
    "rdo_adaround_grad": (C.c_int, [C.POINTER(AdaDesc), P, P, P, P, P, C.c_int, P, P]),
    "rdo_adaround_apply": (C.c_int, [C.POINTER(AdaDesc), P, P, P, P, C.c_float, C.c_float, P, P, P, P, P, P, P, P, P, P, C.c_float, C.c_float, P]),
    "rdo_uaq_fakequant": (C.c_int, [C.POINTER(AdaDesc), P, P, P, P, P, P]),
    "rdo_uaq_init_minmax": (C.c_int, [P, C.c_int32, C.c_int64, C.c_int32, P, P, P]),
    "rdo_actquant_perchannel": (C.c_int, [P, C.c_int64, C.c_int32, C.c_int32, P, P, P]),
    "rdo_actquant_workspace": (C.c_int64, [C.c_int32]),
    "rdo_gather_qdrop": (C.c_int, [P, P, P, P, C.c_int32, C.c_int32, C.c_int64, C.c_float, C.c_uint32, P, P, P]),
    "rdo_lp2_loss_grad": (C.c_int, [P, P, P, P, C.c_int32, C.c_int64, C.c_int32, C.c_float, P, P, P]),
    "rdo_lp_loss_grad": (C.c_int, [P, P, P, P, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_float, P, P, P, P]),
    "rdo_lrelu_fwd": (C.c_int, [P, C.c_int64, P, P]),
    "rdo_lrelu_bwd": (C.c_int, [P, P, C.c_int64, P, P]),
    "rdo_relu_fwd": (C.c_int, [P, C.c_int64, P, P]),
    "rdo_window_attention_fwd": (C.c_int, [C.POINTER(AttnDesc), P, P, P, P, P]),
    "rdo_window_attention_pv": (C.c_int, [C.POINTER(AttnDesc), P, P, P, P]),
    "rdo_window_attention_bwd": (C.c_int, [C.POINTER(AttnDesc), P, P, P, P, P]),
    "rdo_layer_norm_bwd": (C.c_int, [P, P, P, C.c_int64, C.c_int32, C.c_float, P, P, C.c_int32, P]),
    "rdo_add_layer_norm": (C.c_int, [P, P, P, P, C.c_int64, C.c_int32, C.c_float, P, P, P]),
    "rdo_layer_norm_bwd_add": (C.c_int, [P, P, P, P, P, C.c_int64, C.c_int32, C.c_float, P, P, C.c_int32, P]),
    "rdo_add3": (C.c_int, [P, P, P, C.c_int64, P, P]),
    "rdo_linear_h2_supported": (C.c_int, [C.c_int64, C.c_int32, C.c_int32]),
    "rdo_split_h2_linear": (C.c_int, [P, C.c_int32, C.c_int32, C.c_float, P, P]),
    "rdo_linear_h2": (C.c_int, [P, C.c_int64, C.c_int32, C.c_int32, P, C.c_float, P, C.c_int32, P, P]),
    "rdo_linear_h2_epi": (C.c_int, [P, C.c_int64, C.c_int32, C.c_int32, P, C.c_float, P, C.c_int32, C.c_int32, P, P, P, P]),
    "rdo_gelu_fwd": (C.c_int, [P, C.c_int64, P, P]),
    "rdo_gelu_bwd": (C.c_int, [P, P, C.c_int64, P, P]),
    "rdo_round": (C.c_int, [P, C.c_int64, P, P]),
    "rdo_ssim_level": (C.c_int, [P, P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_float, C.c_float, P, P, P]),
    "rdo_avg_pool2": (C.c_int, [P, C.c_int32, C.c_int32, C.c_int32, P, P]),
    "rdo_relu_bwd": (C.c_int, [P, P, C.c_int64, P, P]),
    "rdo_pixel_shuffle": (C.c_int, [P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, P]),
    "rdo_add": (C.c_int, [P, P, C.c_int64, P, P]),
    "rdo_gdn_bwd_t": (C.c_int, [P, P, P, C.c_int64, C.c_int32, P, P]),
    "rdo_gdn_bwd_dx": (C.c_int, [P, P, P, P, C.c_int64, C.c_int32, P, P]),
    "rdo_nchw_to_nhwc": (C.c_int, [P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, P]),
    "rdo_iter_advance": (C.c_int, [P, P]),
    "rdo_zero_insert": (C.c_int, [P] + [C.c_int32] * 9 + [P, P]),
    "rdo_tconv_expand": (C.c_int, [P, P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, P]),
    "rdo_tconv_fold": (C.c_int, [P, P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, P]),
    "rdo_layer_norm": (C.c_int, [P, P, P, C.c_int64, C.c_int32, C.c_float, P, P]),
    "rdo_factorized_likelihood_fwd": (C.c_int, [P, P, P, C.c_int64, C.c_int32, P, P, P]),
    "rdo_factorized_likelihood_bwd": (C.c_int, [P, P, C.c_int64, C.c_int32, C.c_float, P, P]),
    "rdo_gaussian_likelihood_fwd": (C.c_int, [P, P, P, C.c_int64, C.c_float, P, P, P]),
    "rdo_gaussian_likelihood_bwd": (C.c_int, [P, P, P, C.c_int64, C.c_float, C.c_float, P, P, P]),
    "rdo_neg_log2_sum": (C.c_int, [P, C.c_int64, C.c_float, P, P]),
    "rdo_sq_diff_sum": (C.c_int, [P, P, C.c_int64, C.c_float, C.c_int32, P, P]),
    "rdo_h2_overflow": (C.c_int, [C.c_int]),
    "rdo_h2_bind_flag": (C.c_int, [C.c_void_p]),
    "rdo_split_h2": (C.c_int, [P, C.c_int64, C.c_int32, C.c_float, P, P]),
    "rdo_split_h2_conv": (C.c_int, [P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, P, P]),
    "rdo_conv2d_fwd_h2_workspace": (C.c_int64, [C.POINTER(ConvDesc)]),
    "rdo_conv2d_fwd_h2_supported": (C.c_int, [C.POINTER(ConvDesc)]),
    "rdo_conv2d_fwd_h2": (C.c_int, [C.POINTER(ConvDesc), P, C.c_float, P, C.c_float, P, P, P, P, P, P, P, C.c_float, P, C.c_int64, P]),
    "rdo_conv2d_wgrad_h2_supported": (C.c_int, [C.POINTER(ConvDesc)]),
    "rdo_conv2d_wgrad_h2_layer_supported": (C.c_int, [C.POINTER(ConvDesc)]),
    "rdo_conv2d_wgrad_h2": (C.c_int, [C.POINTER(ConvDesc), P, C.c_float, P, C.c_float, P, C.c_int, P]),
    "rdo_conv2d_fwd_h2_tail_supported": (C.c_int, [C.POINTER(ConvDesc)]),
    "rdo_conv2d_fwd_h2_tail": (C.c_int, [C.POINTER(ConvDesc), P, C.c_float, P, C.c_float, P, P, C.c_float, P, P, P, C.c_int32, C.c_float, C.c_int32, P,
                                         C.c_float, P, P]),
    "rdo_conv2d_fwd_ksplit": (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.c_int64]),
    "rdo_conv2d_fwd_partials": (C.c_int, [C.POINTER(ConvDesc), P, P, P, P, C.c_int64, P]),
    "rdo_loss_act_bwd_splitk": (C.c_int, [P, C.c_int32, P, P, P, P, P, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_int32, P, P, P, P, P]),
    "rdo_gather_qdrop_h2": (C.c_int, [P, P, P, P, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_uint32, P, P, C.c_float, P, P]),
    "rdo_loss_act_bwd": (C.c_int, [P, P, P, C.c_float, P, P, P, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_int32, P, P, P, P, C.c_float, P, P]),
    "rdo_loss_gdn_bwd": (C.c_int, [P, P, P, P, P, P, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_int32, P, P, P, P, C.c_float, P, P]),
    "rdo_gdn_bwd_dx_h2": (C.c_int, [P, P, P, P, C.c_int64, C.c_int32, C.c_int32, P, P, C.c_float, P]),
    "rdo_pixel_shuffle_h2": (C.c_int, [P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, P, C.c_float, P]),
    "rdo_pixel_unshuffle2": (C.c_int, [P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, P, C.c_float, P]),
    "rdo_plan_create": (P, []),
    "rdo_plan_destroy": (None, [P]),
    "rdo_plan_begin_record": (C.c_int, [P]),
    "rdo_plan_end_record": (C.c_int, [P]),
    "rdo_plan_suspend_record": (C.c_int, [C.c_int]),
    "rdo_plan_num_ops": (C.c_int, [P]),
    "rdo_plan_run": (C.c_int, [P, C.c_int, C.c_int, P]),
    "rdo_plan_prepare": (C.c_int, [P, C.c_int]),
    "rdo_plan_run_then": (C.c_int, [P, P, C.c_int, P]),
    "rdo_plan_op_info": (C.c_int, [P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rdo_plan_profile": (C.c_int, [P, C.POINTER(C.c_float), P]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib():
    """Load (once) and return the shared library; raise loudly if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` or "
                               f"`make -C rdo-ptq_amd/csrc` (no CPU fallback exists)")
        # PyTorch first: it brings its own copy of the HIP runtime, and the process must hold ONE -- loaded the other way round (this
        # library, then torch: `python __graft_entry__.py smoke` runs build() before smoke()) the library's launches fail with
        # "no ROCm-capable device is detected"
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"librdoptq_hip {what} failed ({rc}): {lib().rdo_last_error().decode()}")
