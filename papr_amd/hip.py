"""ctypes binding of libpapr_hip.so (include/papr_hip.h).

There is deliberately no fallback: if the library is missing or a call fails this raises.
torch is imported first so that the HIP runtime torch already loaded (soname libamdhip64.so.7)
is the one the library binds to -- device pointers and streams are shared with torch.
"""
import ctypes as C
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PAPR_HIP_LIB", os.path.join(_PKG, "libpapr_hip.so"))   # override: instrumented builds (scripts/probes)

ACT = {"none": 0, "relu": 1, "leakyrelu": 2}
EXPECTED_ABI = 27
# `mode` argument of papr_mlp_fwd / papr_mlp_bwd (include/papr_hip.h: PAPR_MLP_*)
MLP_MODES = {"h3": 0, "h1": 1, "f32": 2, "fwd": 3, "dgrad": 4, "layers": 5, "h1_f32rows": 6, "h3_f16rows": 7}
# process-wide A/B switches of the library (papr_set_switch; PAPR_SW_* in include/papr_hip.h).  The library itself reads no environment:
# lib() forwards these historical variable names once, when it loads the library (scripts/probes, tests/test_hip_chain_variants.py)
SWITCHES = {"PAPR_C4_GENERIC": 0, "PAPR_C4_FUSED": 1, "PAPR_C4_EARLY": 2, "PAPR_KNN_BLOCKS": 3, "PAPR_KNN_T": 4, "PAPR_WGRAD_WGS": 5, "PAPR_NT_VARIANT": 6, "PAPR_C4_DMA": 7, "PAPR_TN_JOBPAR": 8, "PAPR_C4_PAIRS": 9, "PAPR_C4_PHASE": 10, "PAPR_C4_SUBPHASE": 11, "PAPR_C4_WCOPIES": 12, "PAPR_TN_TR": 13}

EXPORTS = [
    "papr_abi_version", "papr_last_error", "papr_ray_knn_workspace_bytes", "papr_ray_knn",
    "papr_feature_widths", "papr_build_features_fwd", "papr_build_features_bwd", "papr_build_features_bwd_pairs",
    "papr_segment_reduce_workspace_bytes", "papr_segment_reduce", "papr_group_pairs_workspace_bytes", "papr_group_pairs", "papr_points_knn",
    "papr_rownorm_fwd", "papr_rownorm_bwd", "papr_row_dots", "papr_qk_bias_bwd_workspace_bytes", "papr_qk_bias_bwd", "papr_mse_workspace_bytes", "papr_mse_fwd", "papr_ln_fold_fwd", "papr_ln_fold_bwd", "papr_ln_fold_fwd_batch", "papr_ln_fold_bwd_batch", "papr_mlp_fwd_workspace_bytes", "papr_mlp_bwd_workspace_bytes", "papr_mlp_saved_floats", "papr_mlp_bwd_needs_weight_t", "papr_mlp_fwd",
    "papr_mlp_bwd", "papr_mlp_bwd_takes_f16_rows",
    "papr_attn_tail_fwd", "papr_attn_tail_bwd", "papr_conv3x3_weight_halfs", "papr_conv3x3_workspace_bytes", "papr_conv3x3_fwd", "papr_conv3x3_wgrad_workspace_bytes", "papr_conv3x3_wgrad", "papr_maxpool2_fwd", "papr_maxpool2_bwd", "papr_upconv2x2_fwd", "papr_upconv2x2_dgrad", "papr_upconv2x2_wgrad_workspace_bytes", "papr_upconv2x2_wgrad", "papr_conv1x1_fwd", "papr_conv1x1_bwd_workspace_bytes", "papr_conv1x1_bwd", "papr_small_unet_state_bytes", "papr_small_unet_bwd_workspace_bytes", "papr_small_unet_fwd", "papr_small_unet_bwd", "papr_adam_step", "papr_adam_step_scaled", "papr_composite_fwd", "papr_composite_bwd_workspace_bytes", "papr_composite_bwd", "papr_profile_enable", "papr_profile_collect", "papr_set_switch", "papr_get_switch",
]


class FeatureDesc(C.Structure):
    _fields_ = [("k", C.c_int32), ("feat_dim", C.c_int32), ("L_key", C.c_int32 * 3), ("L_qry", C.c_int32),
                ("L_val", C.c_int32 * 2), ("with_self", C.c_int32), ("key_has_feats", C.c_int32),
                ("val_has_feats", C.c_int32), ("pe_factor", C.c_float), ("pe_mult", C.c_float), ("eps", C.c_float),
                ("ld_key", C.c_int32), ("ld_qry", C.c_int32), ("ld_val", C.c_int32)]


class Layer(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("weight_t", C.c_void_p), ("bias", C.c_void_p),
                ("n_in", C.c_int32), ("n_out", C.c_int32), ("ldw", C.c_int32), ("ldwt", C.c_int32),
                ("n_skip", C.c_int32), ("skip_col", C.c_int32), ("act", C.c_int32)]


class F16Rows(C.Structure):
    """papr_f16_rows: gradient rows in the one-product runs' input format."""
    _fields_ = [("hi", C.c_void_p), ("inv", C.c_void_p), ("scale", C.c_void_p), ("max", C.c_void_p), ("ld", C.c_int32), ("lo", C.c_void_p)]


class RowNorm(C.Structure):
    _fields_ = [("eps", C.c_float), ("width", C.c_int32), ("stats", C.c_void_p),
                ("dot_rows", C.c_void_p), ("ld_dot", C.c_int32), ("rows_per_dot", C.c_int32), ("dots", C.c_void_p),
                ("given_mean", C.c_void_p), ("raw_mean", C.c_void_p), ("leave_input", C.c_int32)]


class ProfileRecord(C.Structure):
    _fields_ = [("kernel", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("M", C.c_int64), ("ms", C.c_float),
                ("bytes", C.c_int64), ("flops", C.c_int64)]


class TailDesc(C.Structure):
    _fields_ = [("k", C.c_int32), ("d_model", C.c_int32), ("C", C.c_int32), ("ld_kp", C.c_int32),
                ("ld_qp", C.c_int32), ("ld_v", C.c_int32), ("score_act", C.c_int32), ("normalize", C.c_int32),
                ("bkg_score", C.c_float), ("scale_dim", C.c_int32), ("precomputed_dots", C.c_int32)]


class LnFoldJob(C.Structure):            # papr_ln_fold_job
    _fields_ = [("w", C.c_void_p), ("n_out", C.c_int32), ("n_in", C.c_int32), ("ldw", C.c_int32), ("c", C.c_void_p), ("a2", C.c_void_p), ("b2", C.c_void_p),
                ("eff_w", C.c_void_p), ("ld_eff", C.c_int32), ("eff_b", C.c_void_p), ("d_eff_w", C.c_void_p), ("d_eff_b", C.c_void_p), ("d_w", C.c_void_p),
                ("d_a2", C.c_void_p), ("d_b2", C.c_void_p)]


LN_FOLD_MAX_JOBS = 8


class UnetDesc(C.Structure):             # papr_unet_desc
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("c_in", C.c_int32), ("n_classes", C.c_int32), ("one_product", C.c_int32),
                ("conv_w", C.c_void_p * 5), ("conv_w_stride", (C.c_int64 * 4) * 5), ("conv_b", C.c_void_p * 5),
                ("up_w", C.c_void_p * 2), ("up_b", C.c_void_p * 2), ("out_w", C.c_void_p), ("out_b", C.c_void_p)]


class UnetGrads(C.Structure):            # papr_unet_grads
    _fields_ = [("conv_w", C.c_void_p * 5), ("conv_b", C.c_void_p * 5), ("up_w", C.c_void_p * 2), ("up_b", C.c_void_p * 2),
                ("out_w", C.c_void_p), ("out_b", C.c_void_p)]


_lib = None


def library_present():
    return os.path.exists(LIB_PATH)


def lib():
    """The loaded library (loads on first use; raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("papr_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the render path)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
    L.papr_abi_version.restype = C.c_int
    if L.papr_abi_version() != EXPECTED_ABI:         # a stale or variant build would be called with shifted arguments
        raise RuntimeError("papr_amd: %s has ABI version %d, this package binds version %d -- rebuild it "
                           "(`python -m papr_amd.build --force`)" % (LIB_PATH, L.papr_abi_version(), EXPECTED_ABI))
    L.papr_last_error.restype = C.c_char_p
    L.papr_ray_knn_workspace_bytes.restype = C.c_size_t
    L.papr_ray_knn_workspace_bytes.argtypes = [i64, i64]
    L.papr_ray_knn.argtypes = [vp, i64, vp, vp, i64, i64, i32, f32, vp, vp, vp, vp]
    L.papr_feature_widths.argtypes = [C.POINTER(FeatureDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.papr_build_features_fwd.argtypes = [C.POINTER(FeatureDesc), vp, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp]
    L.papr_build_features_bwd.argtypes = [C.POINTER(FeatureDesc), vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, vp]
    L.papr_build_features_bwd_pairs.argtypes = [C.POINTER(FeatureDesc), vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp]
    L.papr_segment_reduce.argtypes = [vp, vp, vp, i64, i64, vp, vp, vp, i32, i32, i32, vp, vp, vp, i32, vp, vp]
    L.papr_segment_reduce_workspace_bytes.restype = C.c_size_t
    L.papr_segment_reduce_workspace_bytes.argtypes = [i64]
    L.papr_group_pairs_workspace_bytes.argtypes = [i64, i64]
    L.papr_group_pairs_workspace_bytes.restype = C.c_size_t
    L.papr_group_pairs.argtypes = [vp, i64, i64, i32, vp, vp, vp, vp, C.c_size_t, vp]
    L.papr_points_knn.argtypes = [vp, i64, vp, i64, i32, vp, vp, vp]
    L.papr_rownorm_fwd.argtypes = [vp, i64, i32, i32, f32, vp, vp, vp]
    L.papr_rownorm_bwd.argtypes = [vp, vp, vp, i64, i32, i32, f32, vp, vp]
    L.papr_row_dots.argtypes = [vp, i64, i32, i32, vp, i32, i32, vp, vp]
    L.papr_mse_workspace_bytes.restype = C.c_size_t
    L.papr_mse_workspace_bytes.argtypes = []
    L.papr_mse_fwd.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    L.papr_qk_bias_bwd_workspace_bytes.restype = C.c_size_t
    L.papr_qk_bias_bwd_workspace_bytes.argtypes = [i32]
    L.papr_qk_bias_bwd.argtypes = [vp, i32, i32, i32, vp, i64, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.papr_conv3x3_weight_halfs.restype = C.c_size_t
    L.papr_conv3x3_weight_halfs.argtypes = [i32, i32]
    L.papr_conv3x3_workspace_bytes.restype = C.c_size_t
    L.papr_conv3x3_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.papr_conv3x3_fwd.argtypes = [vp, i32, i32, i32, i32, vp, i64, i64, i64, i64, i32, vp, i32, i32, vp, vp, i32, vp]
    L.papr_conv3x3_wgrad_workspace_bytes.restype = C.c_size_t
    L.papr_conv3x3_wgrad_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.papr_conv3x3_wgrad.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp]
    L.papr_maxpool2_fwd.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp]
    L.papr_maxpool2_bwd.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    L.papr_upconv2x2_fwd.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, vp, vp]
    L.papr_upconv2x2_dgrad.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp, vp]
    L.papr_upconv2x2_wgrad_workspace_bytes.restype = C.c_size_t
    L.papr_upconv2x2_wgrad_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.papr_upconv2x2_wgrad.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    L.papr_conv1x1_fwd.argtypes = [vp, i64, i32, vp, vp, i32, vp, vp]
    L.papr_conv1x1_bwd_workspace_bytes.restype = C.c_size_t
    L.papr_conv1x1_bwd_workspace_bytes.argtypes = [i64, i32, i32]
    L.papr_conv1x1_bwd.argtypes = [vp, vp, i64, i32, vp, i32, vp, vp, vp, vp, vp]
    L.papr_small_unet_state_bytes.restype = C.c_size_t
    L.papr_small_unet_state_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.papr_small_unet_bwd_workspace_bytes.restype = C.c_size_t
    L.papr_small_unet_bwd_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.papr_small_unet_fwd.argtypes = [C.POINTER(UnetDesc), vp, vp, vp, i32, vp]
    L.papr_small_unet_bwd.argtypes = [C.POINTER(UnetDesc), vp, vp, vp, vp, C.POINTER(UnetGrads), vp, vp]
    L.papr_adam_step.argtypes = [vp, i32, vp, i32, vp]
    L.papr_adam_step_scaled.argtypes = [vp, i32, vp, i32, vp, vp, vp]
    L.papr_composite_fwd.argtypes = [vp, vp, i32, i32, vp, i64, i32, i32, vp, vp]
    L.papr_composite_bwd_workspace_bytes.restype = C.c_size_t
    L.papr_composite_bwd_workspace_bytes.argtypes = [i64]
    L.papr_composite_bwd.argtypes = [vp, vp, vp, i32, i32, vp, i64, i32, i32, vp, vp, vp, vp, vp]
    L.papr_ln_fold_fwd.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp]
    L.papr_ln_fold_bwd.argtypes = [vp, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.papr_ln_fold_fwd_batch.argtypes = [C.POINTER(LnFoldJob), i32, vp]
    L.papr_ln_fold_bwd_batch.argtypes = [C.POINTER(LnFoldJob), i32, vp]
    L.papr_mlp_fwd_workspace_bytes.restype = C.c_size_t
    L.papr_mlp_fwd_workspace_bytes.argtypes = [i64]
    L.papr_mlp_bwd_workspace_bytes.restype = C.c_size_t
    L.papr_mlp_bwd_workspace_bytes.argtypes = [i64]
    L.papr_mlp_bwd_needs_weight_t.argtypes = [C.POINTER(Layer), i32, i32, i32]
    L.papr_mlp_saved_floats.restype = C.c_size_t
    L.papr_mlp_saved_floats.argtypes = [i32, i64]
    L.papr_mlp_fwd.argtypes = [C.POINTER(Layer), i32, vp, i32, i64, C.POINTER(vp), C.POINTER(C.c_int32), vp, C.POINTER(RowNorm), C.POINTER(RowNorm), vp, i32, vp]
    L.papr_mlp_bwd.argtypes = [C.POINTER(Layer), i32, vp, i32, i64, C.POINTER(vp), C.POINTER(C.c_int32), vp, vp, C.POINTER(F16Rows), vp, vp,
                               i32, C.POINTER(vp), C.POINTER(vp), vp, vp, i32, vp]
    L.papr_mlp_bwd_takes_f16_rows.argtypes = [C.POINTER(Layer), i32, C.POINTER(C.c_int32), i32, i32]
    L.papr_attn_tail_fwd.argtypes = [C.POINTER(TailDesc), vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp]
    L.papr_attn_tail_bwd.argtypes = [C.POINTER(TailDesc), vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                     C.POINTER(F16Rows), C.POINTER(F16Rows), vp]
    L.papr_profile_enable.argtypes = [i32]
    L.papr_profile_collect.argtypes = [C.POINTER(ProfileRecord), i32]
    L.papr_set_switch.argtypes = [i32, i32]
    L.papr_get_switch.argtypes = [i32]
    for name in EXPORTS:
        getattr(L, name)  # every declared entry point must resolve
    _check_single_hip_runtime()
    for env, which in SWITCHES.items():
        if os.environ.get(env, "") != "":
            if L.papr_set_switch(which, int(os.environ[env])) != 0:
                raise RuntimeError("papr_amd: papr_set_switch(%s) failed: %s" % (env, L.papr_last_error().decode()))
    _lib = L
    return L


def set_switch(name, value):
    """Set a process-wide A/B switch of the library by its historical environment name (SWITCHES); returns the previous value."""
    L = lib()
    old = L.papr_get_switch(SWITCHES[name])
    check(L.papr_set_switch(SWITCHES[name], int(value)), "papr_set_switch")
    return old


def _check_single_hip_runtime():
    seen = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    seen.add(line.split()[-1])
    except OSError:
        return
    if len(seen) > 1:
        raise RuntimeError("papr_amd: two HIP runtimes are mapped (%s); device pointers would not be shared" % sorted(seen))


def profile_enable(on):
    lib().papr_profile_enable(1 if on else 0)


def profile_collect(cap=65536):
    """[(kernel, M, N, K, ms), ...] of the launches timed since the last call (fused-run launches, kernels 9 and 10:
    (kernel, M, n_layers, K0, ms, bytes, flops); batched weight-gradient launches, kernel 8: (kernel, M, n_jobs, 0, ms, bytes, flops); 3x3 convolution 11 and its weight gradient 12: (kernel, pixels, c_out, 9 c_in, ms, bytes, flops))."""
    buf = (ProfileRecord * cap)()
    n = lib().papr_profile_collect(buf, cap)
    return [(r.kernel, r.M, r.N, r.K, r.ms) + ((r.bytes, r.flops) if r.kernel in (8, 9, 10, 11, 12) else ()) for r in buf[:min(n, cap)]]


DEBUG_SYNC = False      # (debugging aid: name every library call on stderr and wait for it -- the last name printed before a GPU fault is the call that faulted;
                        #  train.py switches it on from step PAPR_DEBUG_SYNC_FROM)


def check(code, what):
    if code != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, code, lib().papr_last_error().decode()))
    if DEBUG_SYNC:
        import sys
        sys.stderr.write("papr call: %s\n" % what)
        sys.stderr.flush()
        torch.cuda.synchronize()


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous fp32/int32 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "papr_amd.hip.ptr: need a contiguous device tensor"
    return C.c_void_p(t.data_ptr())


def ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def i32_array(vals):
    return (C.c_int32 * len(vals))(*vals)
