"""ctypes binding of libpcfa_hip.so (C-ABI declared in include/pcfa_hip.h).

There is no CPU fallback: if the shared library is missing or does not match
the ABI version this module raises, and every op in :mod:`pcfa_amd.hip_ops`
raises when handed a non-GPU tensor.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_uint, c_void_p, POINTER

ABI_VERSION = 2
LIB_PATH = os.environ.get("PCFA_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                          "libpcfa_hip.so")  # PCFA_HIP_LIB: A/B builds of the same ABI

PCFA_LOSS = {"aee": 0, "mse": 1, "cosim": 2}


class HipLibraryError(RuntimeError):
    pass


_P = c_void_p  # device pointer
_S4 = POINTER(c_longlong)  # long long[4]

# name -> (restype, argtypes); kept in lock-step with include/pcfa_hip.h
SIGNATURES = {
    "pcfa_abi_version": (c_int, []),
    "pcfa_timing_arm": (c_int, [_P, _P, c_int]),
    "pcfa_null_launch": (c_int, [_P]),
    "pcfa_poison_lds": (c_int, [c_uint, _P]),
    "pcfa_peek_lds": (c_int, [_P, c_int, _P]),
    "pcfa_calib_mfma_f32": (c_longlong, [_P, c_int, c_int, _P]),
    "pcfa_calib_copy": (c_int, [_P, _P, c_longlong, _P]),
    "pcfa_status_string": (c_char_p, [c_int]),
    "pcfa_corr_slab_floats": (c_longlong, [c_int, c_int, c_int]),
    "pcfa_corr_level_offset": (c_longlong, [c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "pcfa_corr_tiled_index": (c_longlong, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "pcfa_corr_f2ext_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_corr_pyramid_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_corr_pyramid_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "pcfa_corr_pyramid_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_corr_pyramid_bwd_windows_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "pcfa_corr_pyramid_bwd_windows": (c_int, [_P, _P, _P, _P, _P, _P, c_size_t, POINTER(c_void_p), c_int, c_int, c_int,
                                              c_int, c_int, c_int, c_int, _P]),
    "pcfa_corr_lookup_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_corr_lookup_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_gemm_f32_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "pcfa_gemm_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_longlong, c_longlong, c_longlong, c_int, c_int, c_int,
                              c_longlong, c_longlong, c_longlong, c_float, c_int, _P, c_size_t, _P]),
    "pcfa_softmax_rows_fwd": (c_int, [_P, _P, c_longlong, c_int, _P]),
    "pcfa_softmax_rows_bwd": (c_int, [_P, _P, _P, c_longlong, c_int, _P]),
    "pcfa_lookup_convc1_packed_floats": (c_longlong, [c_int]),
    "pcfa_lookup_convc1_pack_weights": (c_int, [_P, _P, c_int, c_int, _P]),
    "pcfa_lookup_convc1_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_lookup_convc1_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_spatial_corr_out_size": (c_int, [c_int] * 10 + [POINTER(c_int), POINTER(c_int)]),
    "pcfa_spatial_corr_fwd": (c_int, [_P, _P, _P] + [c_int] * 16 + [_P]),
    "pcfa_spatial_corr_bwd": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 16 + [_P]),
    "pcfa_cost_volume9_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    "pcfa_cost_volume9_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    "pcfa_flownet_corr_out_size": (c_int, [c_int] * 7 + [POINTER(c_int)] * 3),
    "pcfa_flownet_corr_fwd": (c_int, [_P, _P, _P] + [c_int] * 9 + [_P]),
    "pcfa_flownet_corr_bwd": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 9 + [_P]),
    "pcfa_resample2d_fwd": (c_int, [_P, _P, _P] + [c_int] * 8 + [_P]),
    "pcfa_resample2d_bwd": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 8 + [_P]),
    "pcfa_channelnorm_fwd": (c_int, [_P, _P, c_int, c_int, c_longlong, c_int, _P]),
    "pcfa_channelnorm_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_longlong, c_int, _P]),
    "pcfa_box_transform_fwd": (c_int, [_P, _P, _P, c_int, c_longlong, c_int, c_double, c_float, _P]),
    "pcfa_box_transform_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_longlong, c_int, c_double, c_float, _P]),
    "pcfa_pm1_pair_fwd": (c_int, [_P, _P, _P, _P, c_int, c_longlong, _P]),
    "pcfa_pm1_pair_bwd": (c_int, [_P, _P, _P, _P, c_int, c_longlong, _P]),
    "pcfa_extract_deltas_fwd": (c_int, [_P, _P, _P, c_longlong, c_int, c_double, _P]),
    "pcfa_extract_deltas_bwd": (c_int, [_P, _P, _P, c_longlong, c_int, c_double, _P]),
    "pcfa_extract_deltas_joint_fwd": (c_int, [_P, _P, _P, _P, c_longlong, _P]),
    "pcfa_extract_deltas_joint_bwd": (c_int, [_P, _P, _P, _P, _P, c_longlong, _P]),
    "pcfa_flow_loss_workspace_bytes": (c_size_t, []),
    "pcfa_flow_loss_fwd": (c_int, [_P, _S4, _P, _S4, c_int, c_int, c_int, _P, c_longlong, _P, c_longlong,
                                   c_float, c_float, c_int, _P, _P, _P]),
    "pcfa_flow_loss_bwd": (c_int, [_P, _S4, _P, _S4, c_int, c_int, c_int, _P, c_longlong, _P, c_longlong,
                                   c_float, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "pcfa_gru_gates_fwd": (c_int, [_P] * 10 + [c_longlong, c_int, c_int, _P]),
    "pcfa_gru_gates_bwd": (c_int, [_P] * 8 + [c_longlong, _P]),
    "pcfa_gru_update_fwd": (c_int, [_P] * 7 + [c_longlong, c_int, c_int, _P]),
    "pcfa_gru_update_bwd": (c_int, [_P] * 7 + [c_longlong, _P]),
    "pcfa_sepconv5_packed_floats": (c_longlong, [c_int, c_int]),
    "pcfa_sepconv5_algo": (c_int, [c_int]),
    "pcfa_sepconv5_uses_winograd": (c_int, [c_int] * 7),
    "pcfa_sepconv5_pack_weights": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "pcfa_sepconv5_fwd": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_sepconv5_fwd_split": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int,
                                        c_int, _P]),
    "pcfa_gru_gates_bwd_acc": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, _P]),
    "pcfa_sepconv5_fwd_split_masked": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, c_int, _P, c_int, _P, c_int, c_int,
                                               c_int, c_int, c_int, c_int, _P]),
    "pcfa_sepconv5_gru_gates_fwd": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "pcfa_sepconv5_gru_update_fwd": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "pcfa_sepconv5_gru_gates_bwd": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int,
                                            c_int, c_int, _P]),
    "pcfa_sepconv5_gru_update_bwd": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int,
                                             c_int, _P]),
    "pcfa_convex_upsample_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "pcfa_convex_upsample_workspace_floats": (c_longlong, [c_int, c_int, c_int]),
    "pcfa_convex_upsample_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "pcfa_flow_step": (c_int, [_P, _P, _P, _P, _P, c_longlong, _P]),
    "pcfa_conv3x3_act_fwd_pair": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int,
                                          c_float, _P]),
    "pcfa_relu_bwd2": (c_int, [_P, _P, _P, _P, _P, c_longlong, _P]),
    "pcfa_conv3x3_fused_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_conv3x3_masked_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_conv3x3_algo": (c_int, [c_int] * 5),
    "pcfa_conv3x3_workspace_bytes": (c_size_t, [c_int] * 5),
    "pcfa_conv3x3_run": (c_int, [_P] * 6 + [c_int] * 6 + [c_float, c_int, _P, c_size_t, _P]),
    "pcfa_conv3x3_packed_floats": (c_longlong, [c_int, c_int]),
    "pcfa_conv3x3_pack_weights": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "pcfa_conv3x3_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_conv3x3_act_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "pcfa_leaky_relu_bwd": (c_int, [_P, _P, _P, c_float, c_longlong, _P]),
    "pcfa_lbfgs_gram_state_bytes": (c_size_t, [c_int]),
    "pcfa_lbfgs_gram_workspace_bytes": (c_size_t, [c_int, c_longlong]),
    "pcfa_lbfgs_gram_reset": (c_int, [_P, c_int, _P]),
    "pcfa_lbfgs_gram_update": (c_int, [_P, _P, _P, c_float, _P, _P, _P, _P, c_int, c_longlong, _P]),
    "pcfa_lbfgs_gram_direction": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_longlong, _P]),
    "pcfa_lbfgs_workspace_floats": (c_size_t, []),
    "pcfa_lbfgs_pair": (c_int, [_P, _P, _P, c_float, _P, _P, _P, _P, c_int, c_longlong, _P]),
    "pcfa_lbfgs_direction": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_longlong, c_longlong,
                                     _P]),
    "pcfa_bias_relu_fwd": (c_int, [_P, _P, _P, c_longlong, c_int, c_int, _P]),
    "pcfa_relu_bwd": (c_int, [_P, _P, _P, c_longlong, _P]),
    "pcfa_conv_fewin_packed_floats": (c_longlong, [c_int, c_int, c_int]),
    "pcfa_conv_s2_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "pcfa_conv_s2_packed_floats": (c_longlong, [c_int, c_int, c_int]),
    "pcfa_conv_s2_pack": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "pcfa_conv_s2_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "pcfa_conv_s2_bwd_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "pcfa_conv_s2_bwd_packed_floats": (c_longlong, [c_int, c_int, c_int]),
    "pcfa_conv_s2_bwd_pack": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "pcfa_conv_s2_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_conv_s2_ds_packed_floats": (c_longlong, [c_int, c_int]),
    "pcfa_conv_s2_ds_pack": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "pcfa_conv_s2_ds_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "pcfa_conv_s2_ds_bwd_packed_floats": (c_longlong, [c_int, c_int]),
    "pcfa_conv_s2_ds_bwd_pack": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "pcfa_conv_s2_ds_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_conv_fewin_pack": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "pcfa_conv_fewin_packed_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_conv_fewin_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_pwc_warp_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    "pcfa_pwc_warp_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    "pcfa_pwc_warp_bwd_det_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "pcfa_pwc_warp_bwd_det": (c_int, [_P, _P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    "pcfa_conv3x3_fewout_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "pcfa_conv3x3_fewout_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_conv3x3_fewout_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_deconv4s2_fewout_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "pcfa_deconv4s2_fewout_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_deconv4s2_fewout_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pcfa_upsample_bilinear_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "pcfa_upsample_bilinear_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "pcfa_instnorm_workspace_bytes": (c_size_t, [c_int, c_longlong]),
    "pcfa_instnorm_fwd": (c_int, [_P, _P, _P, _P, c_int, c_longlong, c_float, c_int, _P]),
    "pcfa_instnorm_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_longlong, c_int, _P]),
    "pcfa_add_relu_fwd": (c_int, [_P, _P, _P, c_longlong, _P]),
    "pcfa_sum_n": (c_int, [POINTER(c_void_p), c_int, _P, c_longlong, _P]),
    "pcfa_avg_epe": (c_int, [_P, _S4, _P, _S4, c_int, c_int, c_int, _P, _P, _P]),
    "pcfa_sum_squares": (c_int, [_P, c_longlong, _P, _P, _P]),
}

_lib = None


def load(path=None):
    """Load (once) and return the ctypes handle; raises HipLibraryError if unavailable."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise HipLibraryError(
            "libpcfa_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pcfa_amd/csrc`. pcfa_amd has no CPU fallback." % p)
    try:
        lib = ctypes.CDLL(p)
    except OSError as e:  # e.g. missing libamdhip64
        raise HipLibraryError("cannot load %s: %s" % (p, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HipLibraryError("%s does not export %s (stale build?)" % (p, name))
        fn.restype = res
        fn.argtypes = args
    v = lib.pcfa_abi_version()
    if v != ABI_VERSION:
        raise HipLibraryError("ABI version mismatch: library %d, binding %d" % (v, ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().pcfa_status_string(status)
        raise RuntimeError("%s failed: %s (status %d)" % (what, msg.decode() if msg else "?", status))


def strides4(t):
    """long long[4] element strides (b, c, h, w) of a 4-D tensor view."""
    return (c_longlong * 4)(*t.stride())
