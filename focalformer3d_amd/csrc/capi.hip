// Library-level entry points of libff3d_hip.so.
#include "ff3d_common.h"

extern "C" int ff3d_version(void) { return 211; }  // major = version / 100: 2 since round 3 changed entry-point signatures; minor 1 = + the two fused linear entry points, 2 = + the grouped (multi-member) conv / split entry points, 3 = + ff3d_linear_rows (row-owning linear, split-fp16 | bf16), 4 = + ff3d_ffn_rows (the feed-forward step of a decoder layer in one launch), 5 = + ff3d_conv3x3_halo_f16x3_tiled, ff3d_conv3x3_small_f16x3_tiled (tiled weight planes), 6 = + ff3d_local_attention_pair (+ _workspace_halfs), ff3d_msda_gather_rows, 7 = + ff3d_linear_wgrad_f16x3 (+ ff3d_absmax_partials_f32, ff3d_linear_wgrad_slices), 8 = + ff3d_heatmap_box_gather, ff3d_box_class_mask (the heatmap_box branch), 9 = + ff3d_conv3x3_halo_f16x3_nchwsrc

extern "C" const char* ff3d_status_string(int status) {
  switch (status) {
    case FF3D_OK: return "ok";
    case FF3D_ERR_BAD_SHAPE: return "bad shape (a size is <= 0 or exceeds a documented limit)";
    case FF3D_ERR_BAD_DTYPE: return "unknown dtype code";
    case FF3D_ERR_ALIGNMENT: return "pointer not 16-byte aligned";
    case FF3D_ERR_NULL: return "required pointer is NULL";
    case FF3D_ERR_UNSUPPORTED: return "configuration not supported by this build";
    case FF3D_ERR_LAUNCH: return "HIP launch failed";
    default: return "unknown status";
  }
}
