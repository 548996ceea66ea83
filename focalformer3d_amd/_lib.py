"""ctypes binding of the C ABI declared in include/ff3d.h.

There is no CPU or eager fallback: if libff3d_hip.so cannot be loaded, every op raises.
"""
import ctypes as C
import os

# PyTorch-ROCm ships its own libamdhip64.so (same SONAME as /opt/rocm's).  It must be the HIP runtime of
# the process - streams and device pointers handed to the kernels come from it - so torch is imported
# (and its runtime mapped) BEFORE libff3d_hip.so resolves its libamdhip64.so.7 dependency.  Loading the
# library first binds it to /opt/rocm's copy and every launch on a torch stream then fails.
import torch  # noqa: F401  (load order matters, see above)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FF3D_LIB', os.path.join(_PKG, 'lib', 'libff3d_hip.so'))

_vp, _i, _i64, _f, _u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32
ABI_MAJOR = 2                     # ff3d_version() / 100 of the library this table matches


class Scale(C.Structure):
    """ff3d_scale_t (include/ff3d.h, RANGE NORMALISATION): device scalars of one split-fp16 launch."""
    _fields_ = [('a_exp', _vp), ('a2_exp', _vp), ('w_exp', _vp), ('w_bound', _vp), ('res_exp', _vp), ('out_exp', _vp)]


_sp = C.POINTER(Scale)

# name -> (restype, argtypes); must list exactly the symbols include/ff3d.h declares
SIGNATURES = {
    'ff3d_version': (_i, []),
    'ff3d_status_string': (C.c_char_p, [_i]),
    'ff3d_msda_fwd': (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'ff3d_msda_fwd_dev': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ff3d_msda_fused_fwd': (_i, [_vp, _i, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'ff3d_msda_gather_rows': (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'ff3d_self_attention': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _f, _vp]),
    'ff3d_self_attention_f16x3': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _f, _vp]),
    'ff3d_mha_train_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _f, _vp]),
    'ff3d_mha_train_bwd': (_i, [_vp] * 5 + [_f] + [_vp] * 7 + [_i, _i, _i, _i] + [_i64] * 8 + [_f, _vp]),
    'ff3d_absmax_partials_f32': (_i, [_vp, _i64, _vp, _vp]),
    'ff3d_linear_wgrad_slices': (_i, [_i, _i, _i]),
    'ff3d_linear_wgrad_f16x3': (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'ff3d_add_layer_norm': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _vp]),
    'ff3d_sum_add_layer_norm': (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _vp]),
    'ff3d_bias_relu': (_i, [_vp, _vp, _i, _i, _i, _f, _vp]),
    'ff3d_relu_conv3x3_small': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'ff3d_heatmap_nms': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _u32, _vp]),
    'ff3d_topk_workspace_bytes': (C.c_size_t, [_i, _i]),
    'ff3d_topk': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'ff3d_query_gather': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp,
                               _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _u32, _vp]),
    'ff3d_heatmap_box_gather': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ff3d_box_class_mask': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _f, _f, _f, _i, _u32, _vp]),
    'ff3d_bev_flatten': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ff3d_bev_flatten_multi': (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ff3d_sine_embed': (_i, [_vp, _vp, _vp, _i64, _f, _f, _vp]),
    'ff3d_roi_grid_sample': (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _f, _vp, _vp, _i, _vp, _vp]),
    'ff3d_roi_grid_sample_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _f, _vp, _vp, _i, _vp]),
    'ff3d_box_decode': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                             _i, _i, _i, _i, _vp, _vp, _f, _vp]),
    'ff3d_box_update': (_i, [_vp] * 12 + [_i, _i, _i, _i, _i64, _i, _vp, _i, _f, _f, _vp]),
    'ff3d_box_update_rows': (_i, [_vp] * 12 + [_i, _i, _i, _i, _i64, _i, _vp, _i, _f, _f, _vp]),
    'ff3d_pack_detections': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'ff3d_locatt_similar': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'ff3d_locatt_weighting': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'ff3d_locatt_ck2c_loc': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'ff3d_local_attention': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'ff3d_local_attention_pair_workspace_halfs': (_i64, [_i, _i, _i, _i]),
    'ff3d_local_attention_pair': (_i, [_vp] * 11 + [_i, _i, _i, _i, _i, _f, _vp]),
    'ff3d_bev_pool': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ff3d_bev_pool_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'ff3d_circle_nms': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    'ff3d_rotate_nms': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
    'ff3d_boxes_iou_bev': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'ff3d_boxes_iou3d': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'ff3d_gaussian_heatmap_targets': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _f, _i, _vp]),
    'ff3d_nms_bev': (_i, [_vp, _vp, _f, _i, _i, _vp, _vp, _i, _vp]),
    'ff3d_split_f16': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'ff3d_split_f16_nhwc_group': (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    'ff3d_conv3x3_f16x3': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _sp, _vp]),
    'ff3d_conv3x3_f16x3_splitk': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sp, _vp]),
    'ff3d_gemm_f16x3': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _sp, _vp]),
    'ff3d_gemm_f16x3_rowbias': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sp, _vp]),
    'ff3d_conv3x3_f16x3_split_out': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _sp, _vp]),
    'ff3d_conv3x3_halo_f16x3': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _sp, _vp]),
    'ff3d_conv3x3_halo_f16x3_nhwc': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _sp, _vp]),
    'ff3d_conv3x3_halo_f16x3_tiled': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _sp, _vp]),
    'ff3d_conv3x3_halo_f16x3_nchwsrc': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _sp, _vp]),
    'ff3d_conv3x3_small_f16x3': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _sp, _vp]),
    'ff3d_conv3x3_small_f16x3_tiled': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _sp, _vp]),
    'ff3d_conv3x3_halo_f16x3_group': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'ff3d_conv3x3_small_f16x3_group': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'ff3d_msda_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'ff3d_gemm_f16x3_fused': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sp, _vp]),
    'ff3d_linear_f16x3': (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _i, _vp, _i64, _i, _i, _i, _vp]),
    'ff3d_linear_dual_f16x3': (_i, [_vp, _vp, _i, _i64, _vp, _vp, _vp, _vp, _i, _vp, _i64, _i, _i, _i, _vp]),
    'ff3d_linear_kslices_f16x3': (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    'ff3d_linear_add_ln_f16x3': (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'ff3d_linear_rows': (_i, [_vp, _vp, _i, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    'ff3d_ffn_rows': (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _vp]),
    'ff3d_gemm_bf16': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'ff3d_dwconv3x3_pair': (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _sp, _vp]),
    'ff3d_unsplit_f16': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'ff3d_lss_cells': (_i, [_vp] * 9 + [_i] * 5 + [_vp] * 5),
    'ff3d_lss_splat': (_i, [_vp, _i64, _vp, _i, _vp, _vp, _vp, _i, _i, _vp]),
    'ff3d_nchw_to_nhwc': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'ff3d_cam_sample': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
}

_lib = None


def load():
    """Load libff3d_hip.so and bind every entry point; raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'libff3d_hip.so not found at {LIB_PATH}: build it with `python -m focalformer3d_amd.build` '
            '(there is no CPU / eager fallback for the HIP decoder path)')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError -> a declared symbol is missing
        fn.restype, fn.argtypes = res, args
    # ABI major version (ff3d_version() / 100) must be the one this binding was written against: entry points of 1xx libraries
    # take different argument lists (a stale or foreign FF3D_LIB would be called with mismatched arguments: memory faults)
    ver = lib.ff3d_version()
    if ver // 100 != ABI_MAJOR:
        raise RuntimeError(f'{LIB_PATH}: ABI version {ver} (major {ver // 100}), this package binds major {ABI_MAJOR}: rebuild '
                           'with `python -m focalformer3d_amd.build --force`')
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().ff3d_status_string(status).decode()
        raise RuntimeError(f'{what} failed: {msg} (status {status})')
