/*
 * ff3d.h - C ABI of libff3d_hip.so: the MI355X (gfx950) kernels of the FocalFormer3D
 * Hard-Instance-Probing decoder hot path.
 *
 * This is the drop-in operator boundary.  Every entry point replaces one operator (or one
 * fixed group of ATen calls) that the reference head reaches from Python; the reference
 * interface each one stands in for is cited as file:line relative to the reference tree
 * (FD = projects/mmdet3d_plugin/models/dense_heads/focal_decoder.py,
 *  EU = projects/mmdet3d_plugin/models/utils/encoder_utils.py,
 *  UT = projects/mmdet3d_plugin/models/utils/utils.py,
 *  BC = projects/mmdet3d_plugin/core/bbox/coders/transfusion_bbox_coder.py).
 *
 * Conventions
 *  - plain pointers and sizes only; no torch / pybind types.  All `const float*` / `float*`
 *    arguments are DEVICE pointers unless the name ends in `_host`.
 *  - the caller owns every buffer (inputs, outputs, workspaces); the library never allocates
 *    device memory and never synchronises.  Work is enqueued on `stream` (a hipStream_t passed
 *    as void*; NULL = the default stream), so calls are legal inside hipGraph capture.
 *  - tensors are dense row-major in the shape written next to them; fp32 unless noted.
 *  - return value: FF3D_OK (0) or a negative ff3d_status; nothing is enqueued on error.
 *  - stateless and re-entrant.
 */
#ifndef FF3D_H_
#define FF3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ff3d_stream_t; /* hipStream_t */

typedef enum {
  FF3D_OK = 0,
  FF3D_ERR_BAD_SHAPE = -1,   /* a size is <= 0 or exceeds a documented limit          */
  FF3D_ERR_BAD_DTYPE = -2,   /* unknown dtype code                                      */
  FF3D_ERR_ALIGNMENT = -3,   /* a pointer is not aligned as documented (16 B)           */
  FF3D_ERR_NULL = -4,        /* a required pointer is NULL                              */
  FF3D_ERR_UNSUPPORTED = -5, /* legal in the reference, not implemented here (see msg)  */
  FF3D_ERR_LAUNCH = -6       /* hipGetLastError() != hipSuccess after the launch        */
} ff3d_status;

enum { FF3D_F32 = 0, FF3D_BF16 = 1, FF3D_F16_SPLIT = 2 /* two fp16 planes: hi = fp16(x), then lo' = fp16((x - hi) * 2048) */ };
enum { FF3D_MAX_LEVELS = 8, FF3D_HIST_BINS = 4096, FF3D_SPLIT_HINT_INTS = 65 * 64 };

int ff3d_version(void);
const char* ff3d_status_string(int status);

/* ---------------------------------------------------------------------------------------
 * Multi-scale deformable attention, forward.
 * Replaces mmcv-full 1.3.18 `ext_module.ms_deform_attn_forward(value, spatial_shapes,
 * level_start_index, sampling_locations, attention_weights, im2col_step)` reached from
 * FD:927-933 through DeformableDetrTransformerDecoder -> MultiScaleDeformableAttention.
 *   value    (B, Nv, heads, Dh)  fp32 or bf16 (value_dtype), 16-byte aligned
 *   loc      (B, Nq, heads, L, P, 2)  normalised (x, y) in [0,1]
 *   attn_w   (B, Nq, heads, L, P)     softmax-ed over L*P
 *   out      (B, Nq, heads*Dh)        fp32
 *   level_hw_host  L pairs (H_l, W_l) on the HOST; level l starts at sum_{j<l} H_j*W_j and
 *                  the sum over levels must equal Nv.
 * Semantics: out[b,q,h,:] = sum_{l,p} w * bilinear(value_l[b,:,h,:], (x*W_l-0.5, y*H_l-0.5)),
 * zero padding, each corner bounds-checked (align_corners=False).  fp32: Dh % 4 == 0 with Dh/4 a
 * power of two <= 64 (16-byte loads), or Dh itself a power of two <= 64 (scalar loads, tiny
 * heads); bf16: Dh % 8 == 0 with Dh/8 a power of two <= 64.  L <= FF3D_MAX_LEVELS, L*P <= 64.
 */
int ff3d_msda_fwd(const void* value, int value_dtype, const float* loc, const float* attn_w, float* out,
                  int B, int Nv, int Nq, int heads, int Dh, int L, int P, const int32_t* level_hw_host,
                  ff3d_stream_t stream);

/* The same operator with mmcv's own argument convention: `spatial_shapes` (L, 2) int64 (H_l, W_l) and
 * `level_start_index` (L) int64 are DEVICE tensors, exactly what `ext_module.ms_deform_attn_forward(value,
 * spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step)` receives from
 * MultiScaleDeformableAttnFunction.apply (built at FD:837-841 with device='cuda').  The kernel stages the level table
 * from device memory, so a binding needs no `.tolist()` / host synchronisation and the call is graph-capturable.
 * The caller guarantees sum_l H_l*W_l == Nv (it cannot be checked without reading device memory). */
int ff3d_msda_fwd_dev(const void* value, int value_dtype, const int64_t* spatial_shapes_dev,
                      const int64_t* level_start_index_dev, const float* loc, const float* attn_w, float* out, int B,
                      int Nv, int Nq, int heads, int Dh, int L, int P, ff3d_stream_t stream);

/* Same op with the two elementwise prologues of mmcv MultiScaleDeformableAttention.forward
 * fused in: softmax over the L*P logits and loc = ref + off / (W_l, H_l).
 *   value_ld elements between consecutive BEV cells of `value` (0 = heads*Dh, i.e. dense); lets the
 *            value tensors of several decoder layers come from ONE (B, Nv, n_layers*C) GEMM output
 *   ref_pts  (B, Nq, 2)  normalised reference points (valid_ratios == 1, FD:863)
 *   off      rows of heads*L*P*2 raw sampling offsets, row (b*Nq+q) at off + row*off_ld
 *   logits   rows of heads*L*P raw attention logits,   row (b*Nq+q) at logits + row*logits_ld
 * (off and logits may be two column blocks of one GEMM output; *_ld are in elements.) */
int ff3d_msda_fused_fwd(const void* value, int value_dtype, int64_t value_ld, const float* ref_pts, const float* off,
                        int64_t off_ld, const float* logits, int64_t logits_ld, float* out, int B, int Nv, int Nq,
                        int heads, int Dh, int L, int P, const int32_t* level_hw_host, ff3d_stream_t stream);

/* Backward of ff3d_msda_fwd - replaces mmcv `ext_module.ms_deform_attn_backward(value, spatial_shapes,
 * level_start_index, sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc, grad_attn_weight,
 * im2col_step)` behind `MultiScaleDeformableAttnFunction.backward` (training path).  fp32 only.
 *   grad_out (B, Nq, heads*Dh);  grad_value (B, Nv, heads, Dh) ZERO-INITIALISED by the caller (atomic accumulation);
 *   grad_sampling_loc (B, Nq, heads, L, P, 2);  grad_attn_w (B, Nq, heads, L, P). */
int ff3d_msda_bwd(const float* value, const float* sampling_loc, const float* attn_w, const float* grad_out,
                  float* grad_value, float* grad_sampling_loc, float* grad_attn_w, int B, int Nv, int Nq, int heads,
                  int Dh, int L, int P, const int32_t* level_hw_host, ff3d_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Query self-attention core: out = softmax(scale * Q K^T) V per (frame, head), fp32.
 * Replaces the scaled-dot-product step of torch `nn.MultiheadAttention` inside mmcv `MultiheadAttention`
 * (operation 'self_attn' of the decoder layer, FocalFormer3D_L.py:312; reached from FD:927-933).  The
 * in/out projections stay GEMMs in the caller.  No attention mask (inference: attn_masks=None, FD:858).
 *   q, k, v  element (b, n, h, d) at ptr + (b*N + n)*ld + h*Dh + d   (column blocks of GEMM outputs)
 *   out      same addressing with ld_o
 * Dh % 16 == 0 (16..64) runs on v_mfma_f32_16x16x4_f32 and needs 16-byte aligned k / v / out and
 * ld_k, ld_v, ld_o % 4 == 0; other Dh <= 64 use a scalar kernel (test-size models). */
int ff3d_self_attention(const float* q, const float* k, const float* v, float* out, int B, int N, int heads, int Dh,
                        int64_t ld_q, int64_t ld_k, int64_t ld_v, int64_t ld_o, float scale, ff3d_stream_t stream);
/* Training route of the same operation: forward with an additive boolean attention mask (the masks of the
 * ground-truth query groups, FD:849-858: mask (B, N, N) uint8, non-zero = query i may not attend key j, shared by the heads;
 * NULL = none) and attention dropout (keep (B, heads, N, N) uint8 keep-mask drawn by the caller's generator, kept probabilities
 * scaled by keep_scale = 1 / (1 - p); NULL = no dropout), saving lse (B, heads, N) = log-sum-exp of every score row; and its
 * backward -> grad_q / grad_k / grad_v (same addressing as q / k / v with their own row strides).  dsum_workspace: (B, heads, N)
 * floats.  fp32; Dh in {4, 8, 16, 32, 64}.  Replaces the fused SDPA kernels torch dispatches nn.MultiheadAttention to. */
int ff3d_mha_train_fwd(const float* q, const float* k, const float* v, const uint8_t* mask, const uint8_t* keep,
                       float keep_scale, float* out, float* lse, int B, int N, int heads, int Dh, int64_t ld_q, int64_t ld_k,
                       int64_t ld_v, int64_t ld_o, float scale, ff3d_stream_t stream);
int ff3d_mha_train_bwd(const float* q, const float* k, const float* v, const uint8_t* mask, const uint8_t* keep,
                       float keep_scale, const float* out, const float* lse, const float* grad_out, float* grad_q,
                       float* grad_k, float* grad_v, float* dsum_workspace, int B, int N, int heads, int Dh, int64_t ld_q,
                       int64_t ld_k, int64_t ld_v, int64_t ld_o, int64_t ld_go, int64_t ld_gq, int64_t ld_gk, int64_t ld_gv,
                       float scale, ff3d_stream_t stream);
/* Weight gradient of a linear layer on the fp16 matrix cores, fp32-class (round 6; training path, SURVEY.md 8f rank 4 - replaces the
 * fp32 "TN" GEMM + column-sum reduce the framework's autograd runs for every nn.Linear of FD:1166-1311's backward):
 *   dw (N, K) = dy (M, N)^T x (M, K),   db (N) = column sums of dy (db NULL: not wanted).
 * x, dy: row-major fp32 with row strides ldx >= K, ldy >= N (floats; K, N, ldx, ldy % 4 == 0, bases 16-byte aligned).  Both operands
 * are split into (hi, lo') fp16 pairs IN the kernel (no fp16 copy in HBM), scaled by powers of two taken from the tensors' maxima:
 * amax_x / amax_dy = the 256 partial maxima ff3d_absmax_partials_f32 wrote for x / dy (a record may be reused while its tensor is
 * unchanged: one x feeds several layers; any upper bound within a factor 2 of the true maximum is a valid entry).
 * workspace: ff3d_linear_wgrad_slices(M, K, N) * (N * K + N) floats; the row slices are added in order (deterministic). */
int ff3d_absmax_partials_f32(const float* x, int64_t n, float* out256, ff3d_stream_t stream);
int ff3d_linear_wgrad_slices(int M, int K, int N);
int ff3d_linear_wgrad_f16x3(const float* x, int64_t ldx, const float* dy, int64_t ldy, const float* amax_x, const float* amax_dy,
                            int M, int K, int N, float* dw, float* db, float* workspace, ff3d_stream_t stream);
/* Same operation and contract on the fp16 matrix cores with fp32-class accuracy (operands as (hi, lo') fp16 pairs, three
 * MFMA passes, fp32 accumulation - the arithmetic of ff3d_gemm_f16x3); Dh = 16 or 32. */
int ff3d_self_attention_f16x3(const float* q, const float* k, const float* v, float* out, int B, int N, int heads, int Dh,
                              int64_t ld_q, int64_t ld_k, int64_t ld_v, int64_t ld_o, float scale, ff3d_stream_t stream);

/* Fused decoder epilogues.
 * ff3d_add_layer_norm: out = LayerNorm(a + b) * gamma + beta over the last dim (rows x C, C <= 1024; b
 *   nullable) - the residual add + `norm` operation pair of the decoder layer (operation_order
 *   FocalFormer3D_L.py:312-313); when out_pos != NULL also out_pos = out + pos, the `query + query_pos`
 *   input of the following attention (mmcv MultiheadAttention / MultiScaleDeformableAttention).
 * ff3d_bias_relu: x = relu(x + bias[c]) in place on an (N, C, HW) map - the folded
 *   BatchNorm shift + ReLU of mmcv ConvModule (FD:151-162, 204-212); bias nullable; upper > 0 clamps the result
 *   from above (6 = the ReLU6 of the neck's MobileNetV2 blocks), upper <= 0 = plain ReLU. */
int ff3d_add_layer_norm(const float* a, const float* b, const float* gamma, const float* beta, const float* pos,
                        float* out, float* out_pos, int64_t rows, int C, float eps, ff3d_stream_t stream);
/* ff3d_sum_add_layer_norm (ABI 2.11): out = LayerNorm(residual + bias + sum_{s < nparts} parts[row * ld_parts + s * C + c]) * gamma +
 *   beta (and out_pos = out + pos): the second half of a K-sliced projection (ff3d_linear_kslices_f16x3) - the partial columns are
 *   added in slice order (deterministic).  bias / residual / pos nullable; C <= 1024; nparts 1 .. 64; ld_parts >= nparts * C. */
int ff3d_sum_add_layer_norm(const float* parts, int nparts, int64_t ld_parts, const float* bias, const float* residual,
                            const float* gamma, const float* beta, const float* pos, float* out, float* out_pos, int64_t rows,
                            int C, float eps, ff3d_stream_t stream);
int ff3d_bias_relu(float* x, const float* bias, int N, int C, int HW, float upper, ff3d_stream_t stream);

/* Final layer of the heatmap head, fused: out = conv3x3_pad1(relu(x + in_bias[c]), w) + bias, K <= 16 output
 * channels, exact fp32 on v_mfma_f32_16x16x4_f32.  Replaces the BatchNorm shift + ReLU of `heatmap_head.0`
 * (mmcv ConvModule, FD:204-212) and the `heatmap_head.1` Conv2d(C -> num_classes, 3, padding=1) (FD:213-220):
 *   x (B, C, H, W) raw output of the first conv (BatchNorm scale folded into its weights), in_bias (C) nullable,
 *   w (K, C, 3, 3), bias (K) nullable, out (B, K, H, W). */
int ff3d_relu_conv3x3_small(const float* x, const float* in_bias, int apply_relu, const float* w, const float* bias,
                            float* out, int B, int C, int H, int W, int K, ff3d_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Hard-Instance-Probing stage (heatmap -> NMS -> top-k -> gathers -> positive mask).
 */

/* FD:631-634 / 662-666 (sigmoid * accumulated mask), FD:549 (two-heatmap mean when logits_b
 * != NULL), FD:672-685 (local-max NMS, classes in small_class_bits use kernel 1).
 *   logits, logits_b  (B, K, H, W)   raw heatmap-head outputs (logits_b nullable)
 *   mask_in   (B, K, H, W) {0,1} fp32, nullable (= all ones)
 *   mask_next (B, K, H, W) nullable: receives a copy of mask_in (ones if mask_in NULL) - the
 *             buffer ff3d_query_gather later clears the new positives in (FD:782); mask_in is
 *             left untouched and is the reference's `multistage_masks` snapshot (FD:633,668)
 *   heat      (B, K, H, W) post-NMS scores
 *   hist      (B, FF3D_HIST_BINS) uint32: per-sample histogram of the positive scores in
 *             linear bins floor(s*4096); zeroed by this call (memset node on `stream`)
 * nms_kernel in {1, 3}. */
int ff3d_heatmap_nms(const float* logits, const float* logits_b, const float* mask_in, float* mask_next,
                     float* heat, uint32_t* hist, int B, int K, int H, int W, int nms_kernel,
                     uint32_t small_class_bits, ff3d_stream_t stream);

/* FD:688 `torch.topk(heat.view(B,-1), k, largest=True, sorted=False)` / FD:574 argsort[:k].
 * Deterministic: output sorted by score descending, ties by lowest flat index (the reference
 * leaves both implementation-defined).  heat (B, n) >= 0, hist from ff3d_heatmap_nms,
 * idx_out (B, k) int64, 1 <= k <= 4096, k <= n.  workspace: ff3d_topk_workspace_bytes(B, n) bytes, 8-byte aligned
 * (B*n 64-bit candidate keys + one counter per frame; contents need no initialisation).  Two kernels: a many-block
 * candidate compaction over the score rows (threshold bin from the histogram) and a one-block-per-frame sort. */
size_t ff3d_topk_workspace_bytes(int B, int n);
int ff3d_topk(const float* heat, const uint32_t* hist, int64_t* idx_out, void* workspace, int B, int n, int k,
              ff3d_stream_t stream);

/* FD:690-706 (class / cell split, feature gather + class embedding, position and score
 * gathers) fused with FD:725-782 (positive mask scatter + 3x3 dilation + accumulate).
 *   feat   (B, C, H*W)  stage BEV map            heat (B, K, H*W) post-NMS scores
 *   idx    (B, k) int64 flat indices cls*H*W + cell
 *   cls_w  (C, K), cls_b (C)   `class_encoding` Conv1d(K->C, 1) weight / bias (FD:290)
 *   qfeat  element (b, q_offset+j, c) at qfeat + b*qf_sb + (q_offset+j)*qf_sq + c*qf_sc
 *   qpos   (B, Nq, 2) cell centres (x+0.5, y+0.5)      qscore (B, K, Nq)
 *   qlabel (B, Nq) int64
 *   mask   (B, K, H*W) nullable; mask_mode 0 none | 1 'poscls' | 2 'pos' (FD:725-731);
 *          entries covered by the (dilated) positives are set to 0. */
int ff3d_query_gather(const float* feat, const float* heat, const int64_t* idx, const float* cls_w,
                      const float* cls_b, float* qfeat, int64_t qf_sb, int64_t qf_sq, int64_t qf_sc, float* qpos,
                      float* qscore, int64_t* qlabel, float* mask, int B, int C, int K, int H, int W, int k,
                      int q_offset, int Nq, int mask_mode, int nms_kernel, uint32_t small_class_bits,
                      ff3d_stream_t stream);

/* The heatmap_box branch of the stages (FD:231-287 thin form; no shipped config enables it).
 * FD:606-629 / 641-660 (task -> class expansion of the task head's output) + FD:708-722 (cell
 * offsets, clips, gather at the stage's proposals).
 *   raw        (B, T * 10, H, W)  output of the stage's (conv, conv) task head, 10 box values
 *              (reg 2, height 1, dim 3, rot 2, vel 2) per task group
 *   idx        (B, k) int64 flat indices cls*H*W + cell (ff3d_topk)
 *   class_task_host  K int32 on the HOST: class -> task group (FD:232-239)
 *   query_box  (B, 10, Nq): columns [q_offset, q_offset + k) are written. */
int ff3d_heatmap_box_gather(const float* raw, const int64_t* idx, const int32_t* class_task_host, float* query_box,
                            int B, int K, int T, int H, int W, int k, int q_offset, int Nq, ff3d_stream_t stream);

/* FD:732-768, the box part of mask_heatmap_mode = 'boxcls', fused with the dilation and the
 * accumulate of FD:774-782: every BEV cell whose centre lies inside the (shrunk) box of one of the
 * stage's k queries - the FIRST such query, as mmdet3d v0.17.1's points_in_boxes_gpu reports it
 * (FD:742,756) - clears its 3x3 window (1x1 for the classes in small_class_bits) in that query's
 * class plane of `mask`.  Call after ff3d_query_gather(mask_mode 1) on the same mask.
 *   query_box (B, 10, Nq), qlabel (B, Nq) int64: columns [q_offset, q_offset + k) are read, k <= 1024
 *   mask      (B, K, H, W) in place
 *   coder_host  5 floats on the HOST: out_size_factor, voxel_x, voxel_y, pc_range_x, pc_range_y
 *   range_host  4 floats on the HOST: x0, y0, x1, y1 clip of the box centres (FD:746-748)
 *   margin / min_bev_dim / max_bev_dim: FD:749-753 (1.0 / 0.7 / 10.0 in the reference). */
int ff3d_box_class_mask(const float* query_box, const int64_t* qlabel, float* mask, int B, int K, int H, int W, int k,
                        int q_offset, int Nq, const float* coder_host, const float* range_host, float margin,
                        float min_bev_dim, float max_bev_dim, int nms_kernel, uint32_t small_class_bits,
                        ff3d_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * BEV pyramid flatten: FD:823 `cat([f.flatten(2,3) for f in levels], -1)` + the (Nv,B,C)
 * permute of FD:930, written channels-last, with FD:886 (`+ bev_pos_embed`) optionally fused.
 *   levels_host  L device pointers (held in a HOST array), level l is (B, C, H_l, W_l)
 *   pos_embed    (Nv, C) nullable
 *   out_raw      (B, Nv, C) nullable: the pyramid itself (input of the RoI sampler)
 *   out_value    (B, Nv, C) nullable: pyramid + pos_embed (input of value_proj); fp32, or with value_dtype ==
 *                FF3D_F16_SPLIT two fp16 planes (operand of ff3d_gemm_f16x3), each of B*Nv + 1 rows of C: the kernel
 *                fills the first B*Nv rows, the trailing (zero) row is the caller's; with FF3D_BF16 (round 5) ONE bf16 plane of
 *                B*Nv + 1 rows, value rounded to nearest even (operand of ff3d_gemm_bf16: BASELINE configs[4] mode)
 *   level_exp_host  FF3D_F16_SPLIT only, nullable: L device pointers (held in a HOST array) to the int32 bound exponents
 *                of the levels (|level_l| < 2^(e_l+15): the out_exp of the op that produced or split the level), and
 *   pe_exp       the same for pos_embed (NULL: no pos_embed bound needed when pos_embed is NULL);
 *   value_exp    written: exponent of the value pair = max(e_l, e_pe) + 1; raw_exp written: max(e_l) (bound of out_raw,
 *                passed on to ff3d_roi_grid_sample).  level_exp_host == NULL: the pair is written unscaled (e = 0).
 * C % 4 == 0. */
int ff3d_bev_flatten(const float* const* levels_host, const float* pos_embed, float* out_raw, void* out_value,
                     int value_dtype, int B, int C, int L, const int32_t* level_hw_host,
                     const int32_t* const* level_exp_host, const int32_t* pe_exp, int32_t* value_exp, int32_t* raw_exp,
                     ff3d_stream_t stream);
/* The same pass writing up to 4 value tensors (one per decoder stage: value_s = pyramid + pos_embed_s, FD:886 inside the
 * stage loop FD:835-957): the pyramid is read and transposed once.  *_host arguments are HOST arrays of n_values device
 * pointers (pos_embeds_host[v] may be NULL: value = pyramid). */
int ff3d_bev_flatten_multi(const float* const* levels_host, int n_values, const float* const* pos_embeds_host, float* out_raw,
                           void* const* out_values_host, int value_dtype, int B, int C, int L,
                           const int32_t* level_hw_host, const int32_t* const* level_exp_host,
                           const int32_t* const* pe_exps_host, int32_t* const* value_exps_host, int32_t* raw_exp,
                           ff3d_stream_t stream);

/* UT:40-53 `gen_sineembed_for_position` for 2-d positions, with the FD:869 / FD:883 division by
 * the level-0 grid size fused:  r = pos / (W, H);  emb = cat(sincos(2*pi*r_y / dim_t),
 * sincos(2*pi*r_x / dim_t)).   pos (N, 2) -> emb (N, 256);  dim_t (128) device table
 * 10000^(2*(i//2)/128) supplied by the caller (so it is bit-identical to the host's pow). */
int ff3d_sine_embed(const float* pos, const float* dim_t, float* emb, int64_t N, float W, float H,
                    ff3d_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * RoI grid features: FD:891-919 (box decode BC:54-69 with dim*expand, g x g grid in the box
 * frame FD:1655-1664, yaw rotation, normalisation by the hard-coded range FD:903-909,
 * F.grid_sample per pyramid level, concat + permute).
 *   feat_cl    (B, Nv, C) channels-last pyramid (ff3d_bev_flatten out_raw)
 *   query_box  (B, box_dim, Nq) raw head outputs (center2, height1, dim3, rot2[, vel2])
 *   out        (B*Nq, L*C*g*g) fp32, or bf16 / two fp16 planes of B*Nq + 1 rows (the trailing zero row is the caller's)
 *              when out_dtype == FF3D_BF16 / FF3D_F16_SPLIT (layout 1 only); layout 0: column order [level][channel][point] (reference order,
 *              FD:919); layout 1: [level][point][channel] (coalesced; needs roi_mlp.0.weight with
 *              its columns permuted the same way)
 *   grid_out   (B, Nq, g*g, 2) nullable: the normalised sampling grid
 *   coder_host 5 floats: out_size_factor, voxel_x, voxel_y, pc_range_x, pc_range_y (BC:10-22)
 *   range_host 4 floats: x_min, y_min, x_max, y_max of FD:903-906
 *   feat_exp   FF3D_F16_SPLIT only, nullable: device int32 bound exponent of feat_cl (ff3d_bev_flatten raw_exp); bilinear
 *              samples are convex combinations, so the RoI pair is written with the same exponent (NULL: unscaled)
 * C % 4 == 0, g*g <= 256. */
int ff3d_roi_grid_sample(const float* feat_cl, const float* query_box, void* out, int out_dtype, float* grid_out,
                         int B, int Nq, int C, int L, const int32_t* level_hw_host, int g, int box_dim, float expand,
                         const float* coder_host, const float* range_host, int layout, const int32_t* feat_exp,
                         ff3d_stream_t stream);
/* Backward of ff3d_roi_grid_sample with respect to the pyramid - the gradient the reference obtains from autograd through
 * `F.grid_sample` (FD:914-918) on the training path; query_box is detached there (FD:956), so no box
 * gradient exists.  fp32.
 *   grad_out      (B*Nq, L*C*g*g) gradient of the RoI matrix, column order as `layout` (same meaning as in the forward)
 *   grad_feat_cl  (B, Nv, C) ZERO-INITIALISED by the caller (atomic accumulation, order-dependent rounding as in the
 *                 framework's grid_sample backward)
 * g*g <= 256; any C. */
int ff3d_roi_grid_sample_bwd(const float* grad_out, const float* query_box, float* grad_feat_cl, int B, int Nq, int C,
                             int L, const int32_t* level_hw_host, int g, int box_dim, float expand,
                             const float* coder_host, const float* range_host, int layout, ff3d_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Box update of one decoder stage, FD:936-957 (+ the per-key torch.cat over stages, FD:970-987), one launch:
 *   raw   (B, S, Nq)  output of the fused prediction GEMM without its bias (channel blocks as listed below);  bias (S)
 *   ref   (B, Nq, 2)  normalised reference points;  centre = raw + bias + ref * (W, H)  (FD:936, 945)
 *   prev_box (B, 8|10, Nq) nullable: with roi_based_reg, dim[:2] += prev[3:5], rot += prev[6:8] (FD:949-951)
 *   center / height / dim / rot / vel (nullable) / heat: the (B, n, ld) result tensors of the head; this stage's slice is
 *   written at column offset q0 (ld = stages * Nq);  qpos_out (B, Nq, 2) = the new centres (FD:947);
 *   box_out (B, 8|10, Nq) = cat(center, height, dim, rot[, vel]) (FD:952-956).
 *   channel_offsets_host: 6 int32 = first channel of center, height, dim, rot, vel (-1: no velocity head), heatmap in raw. */
int ff3d_box_update(const float* raw, const float* bias, const float* ref, const float* prev_box, float* center,
                    float* height, float* dim, float* rot, float* vel, float* heat, float* qpos_out, float* box_out, int B,
                    int S, int Nq, int K, int64_t ld, int q0, const int32_t* channel_offsets_host, int roi_based_reg,
                    float W, float H, ff3d_stream_t stream);
/* ff3d_box_update_rows (round 5): ff3d_box_update with raw as the (B * Nq, S) ROW-MAJOR output of a query-major GEMM (the prediction
 *   heads' second layer, DU:540-578, on ff3d_linear_f16x3) instead of (B, S, Nq). */
int ff3d_box_update_rows(const float* raw, const float* bias, const float* ref, const float* prev_box, float* center,
                         float* height, float* dim, float* rot, float* vel, float* heat, float* qpos_out, float* box_out,
                         int B, int S, int Nq, int K, int64_t ld, int q0, const int32_t* channel_offsets_host,
                         int roi_based_reg, float W, float H, ff3d_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * get_bboxes: FD:1317-1331 + BC:71-158 (decode, post_center_range filter; the score threshold
 * is applied only when score_threshold != 0, BC:140-141) + the 200-box cap FD:1395-1400.
 * Inputs are the (B, n, ld) prediction tensors; queries q0 .. q0+Nq-1 of the last dim are used.
 *   qscore (B, K, Nq), qlabel (B, Nq) int64
 *   boxes  (B, max_out, 7 + 2*has_vel), scores (B, max_out), labels (B, max_out) int32,
 *   count  (B) int32: number of valid rows per sample.
 * If a sample keeps more than max_out boxes the max_out best by score are returned in
 * descending score order (ties by query index); otherwise kept boxes stay in query order.
 * coder_host: 5 floats as for ff3d_roi_grid_sample; post_center_range_host: 6 floats (BC:129-135), or NULL = BC's
 * `decode(filter=False)`: no range / score test, query q is written to row q (max_out >= Nq required, any Nq; count = Nq) -
 * a NaN or infinite box is kept in place instead of being dropped by a comparison (the training targets index rows by query).
 * With a range, Nq <= 4096 (LDS sort keys of the 200-box cap).
 * Nq <= 4096. */
int ff3d_box_decode(const float* cls, const float* center, const float* height, const float* dim, const float* rot,
                    const float* vel, int64_t ld, int q0, const float* qscore, const int64_t* qlabel, float* boxes,
                    float* scores, int32_t* labels, int32_t* count, int B, int K, int Nq, int max_out,
                    const float* coder_host, const float* post_center_range_host, float score_threshold,
                    ff3d_stream_t stream);

/* Packed detections for the multi-GPU gather (the fixed-shape replacement of mmdet `multi_gpu_test`'s pickled-bytes
 * `collect_results_gpu`, tools/test.py:229-233): boxes (B, M, box_dim <= 9), scores (B, M), labels (B, M) int32,
 * count (B) int32 -> packed (B, M+1, 11) fp32: row 0 = (count, box_dim, 0, ...), rows 1.. = box values zero-padded
 * to 9 columns | score | label.  One all-gather of this tensor carries every rank's detections. */
int ff3d_pack_detections(const float* boxes, const float* scores, const int32_t* labels, const int32_t* count,
                         float* packed, int B, int M, int box_dim, ff3d_stream_t stream);

/* Per-task circle NMS of get_bboxes (FD:1352-1393, test_cfg.nms_type == 'circle') + keep-mask compaction + the
 * 200-box cap, on the padded output of ff3d_box_decode called with max_out = Nq (no cap).  Replaces the host-side
 * mmdet3d `circle_nms(dets[x,y,score], thresh, post_max_size=83)` loop of FD:1361-1367.
 *   boxes (B, M, box_dim), scores (B, M), labels (B, M) int32, count (B) int32   (M <= 2048)
 *   class_task_host (K) int32: task index of every class (FD:1333-1344); task_radius_host (num_tasks): radius,
 *   <= 0 keeps every box of the task.  Outputs padded to max_out as ff3d_box_decode. */
int ff3d_circle_nms(const float* boxes, const float* scores, const int32_t* labels, const int32_t* count,
                    float* out_boxes, float* out_scores, int32_t* out_labels, int32_t* out_count, int B, int M,
                    int box_dim, int max_out, int K, const int32_t* class_task_host, int num_tasks,
                    const float* task_radius_host, int post_max_size, ff3d_stream_t stream);

/* Per-task rotated-IoU NMS of get_bboxes (FD:1369-1383, test_cfg.nms_type neither None nor 'circle'): the reference
 * calls mmdet3d 0.17.1 `nms_gpu(xywhr2xyxyr(boxes.bev), scores, thresh=task['radius'], pre_maxsize, post_max_size)` per
 * task.  Same buffers / outputs as ff3d_circle_nms; box_dim >= 7 with (x, y, z, w, l, h, yaw, ...);
 * task_thresh_host: IoU threshold per task (<= 0 keeps every box of the task). */
int ff3d_rotate_nms(const float* boxes, const float* scores, const int32_t* labels, const int32_t* count,
                    float* out_boxes, float* out_scores, int32_t* out_labels, int32_t* out_count, int B, int M,
                    int box_dim, int max_out, int K, const int32_t* class_task_host, int num_tasks,
                    const float* task_thresh_host, int pre_max_size, int post_max_size, ff3d_stream_t stream);

/* mmdet3d 0.17.1 iou3d ops used by the reference's TTA merging (core/post_processing/merge_augs.py:137, 150).
 *   ff3d_boxes_iou_bev: `boxes_iou_bev(boxes_a (N,5), boxes_b (M,5)) -> (N, M)` rotated BEV IoU, boxes (x1,y1,x2,y2,angle).
 *   ff3d_nms_bev: `nms_gpu(boxes (n,5), scores (n), thresh, pre_maxsize, post_max_size)`: keep (n) int32 receives the kept
 *   ORIGINAL indices in descending score order (ties: lower index), count (1) their number.  n <= 4096; pass a large
 *   pre_max_size / post_max_size for "None". */
int ff3d_boxes_iou_bev(const float* boxes_a, const float* boxes_b, float* out, int N, int M, ff3d_stream_t stream);
int ff3d_nms_bev(const float* boxes, const float* scores, float thresh, int pre_max_size, int post_max_size,
                 int32_t* keep, int32_t* count, int n, ff3d_stream_t stream);
/* 3-D IoU matrix of LiDAR boxes: mmdet3d `BboxOverlaps3D(coordinate='lidar')` (iou calculator of the reference's
 * HungarianAssigner3D, core/bbox/assigners/hungarian_assigner.py:108, 128) = rotated BEV overlap area x height overlap over
 * the union volume.  boxes_a (N, dim_a), boxes_b (M, dim_b): (x, y, z_bottom, dx, dy, dz, yaw, ...), dim >= 7 -> iou (N, M). */
int ff3d_boxes_iou3d(const float* boxes_a, const float* boxes_b, float* iou, int N, int M, int dim_a, int dim_b,
                     ff3d_stream_t stream);

/* Dense heatmap targets of the training loss, FocalDecoder.get_targets_single FD:1133-1158: per ground-truth box the
 * mmdet3d `gaussian_radius((length, width) in cells, gaussian_overlap)` (fp32, op by op as the reference), radius =
 * max(min_radius, int(r)), the truncated centre cell and `draw_heatmap_gaussian` into the plane of the box's class.
 *   gt_boxes (m, box_dim) x, y, z, dx, dy, dz, yaw, ...   gt_labels (m) int64   heatmap (K, H, W) zero-initialised
 *   coder_host 5 floats: out_size_factor, voxel_x, voxel_y, pc_range_x, pc_range_y (train_cfg).  One block per box. */
int ff3d_gaussian_heatmap_targets(const float* gt_boxes, const int64_t* gt_labels, float* heatmap, int m, int box_dim,
                                  int K, int H, int W, const float* coder_host, float gaussian_overlap, int min_radius,
                                  ff3d_stream_t stream);


/* ---------------------------------------------------------------------------------------
 * Camera-projection sampler: EU:194-261 `I2P.forward` without its dense projections.
 *   ff3d_nchw_to_nhwc : (N, C, HW) -> (N, HW, C) transpose (camera FPN maps arrive NCHW).
 *   ff3d_cam_sample   : for every BEV pillar (b, y, x): project its Z height samples into every
 *       camera (EU:210-242), bilinear-sample the NHWC maps (EU:243), masked mean over cameras
 *       (EU:249), then the 1-head attention over the Z samples (EU:252-258) in its folded form:
 *       score_z = qk[b,pillar,:] . f_z (the key bias is softmax-invariant), softmax over valid z,
 *       ctx = sum_z p_z f_z.  The caller applies the folded projections before (qk) and after
 *       (ctx -> output) as dense GEMMs.
 *   img_cl     (B, Ncam, Hi, Wi, Ci)      lidar2img (B, Ncam, 4, 4)
 *   img_aug    (B, Ncam, 4, 4) nullable   qk (B, H*W, Ci)
 *   ctx        (B, H*W, Ci)               valid (B, H*W) uint8: pillar has >= 1 visible sample
 *   range_host 6 floats (EU:210)          input_hw_host 2 floats (img_metas['input_shape'])
 * Ci % 4 == 0, Ci <= 256, Z <= 32, Ncam <= 8. */
int ff3d_nchw_to_nhwc(const float* in, float* out, int N, int C, int HW, ff3d_stream_t stream);
int ff3d_cam_sample(const float* img_cl, const float* lidar2img, const float* img_aug, const float* qk, float* ctx,
                    uint8_t* valid, int B, int Ncam, int Ci, int Hi, int Wi, int H, int W, int Z,
                    const float* range_host, const float* input_hw_host, ff3d_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Local (k x k window) context attention - counterpart of the reference's own CUDA extension
 * projects/mmdet3d_plugin/models/utils/ops/locatt_ops, used by LocalContextAttentionBlock (EU:109-163) in the
 * `iterbev='bevfusion'` neck blocks (FocalFormer3D_LC*.py).  All maps are (B, C, H, W) fp32; kH == kW odd <= 9.
 *   ff3d_locatt_similar   = localattention.similar_forward (similar.cu:3-38, kernels.cuh:4-42 cc2k):
 *       y[b,h,w,k] = sum_c x_ori[b,c,h,w] * x_loc[b,c,h+dy,w+dx], 0 where the window leaves the map; y (B,H,W,kH*kW)
 *   ff3d_locatt_weighting = localattention.weighting_forward (weighting.cu, kernels.cuh:44-80 ck2c_ori):
 *       y[b,c,h,w] = sum_k x_ori[b,c,h+dy,w+dx] * x_weight[b,h,w,k]  (out-of-map taps skipped)
 *   ff3d_local_attention  = EU:158-161 in one launch: similar -> softmax(scale * .) over the window -> weighting
 *       (the (B,H,W,k*k) tensor is never materialised). */
int ff3d_locatt_similar(const float* x_ori, const float* x_loc, float* y, int B, int C, int H, int W, int kH, int kW,
                        ff3d_stream_t stream);
int ff3d_locatt_weighting(const float* x_ori, const float* x_weight, float* y, int B, int C, int H, int W, int kH,
                          int kW, ff3d_stream_t stream);
int ff3d_local_attention(const float* query, const float* key, const float* value, float* out, int B, int C, int H,
                         int W, int kH, int kW, float scale, ff3d_stream_t stream);
/* ff3d_msda_gather_rows (round 6; the opt-in value mode 'gather_first' of the decoder's cross-attention): value_proj is
 *   linear, so sum_k w_k (W v_k + b) = W (sum_k w_k v_k) + b sum_k w_k - the weighted bilinear gather can run on the UN-projected value
 *   (B, Nv, C) fp32 and the projection on the gathered rows.  Same fused prologue as ff3d_msda_fused_fwd (reference point + offset /
 *   (W_l, H_l), softmax over L * P); every (b, q, head) gathers all C channels at its own locations into
 *   `groups` column groups of hpg = heads / groups heads each, group g at column g * (hpg * C + 32):
 *       [head g*hpg's C channels | ... | the group's hpg sums of IN-MAP corner weights x attention weight | zeros up to 32 columns]
 *   (one group per diagonal block of the projection that follows; K of each block = hpg * C + 32, a multiple of 32).
 *   out_ld >= heads * C + 32 * groups, out_ld % 4 == 0; C in {64, 128, 256}.  mmcv MultiScaleDeformableAttention.forward (FD:927-933) with the projection moved behind
 *   the gather; the default mode (project first, the HBM-bound gather of ff3d_msda_fused_fwd) is unchanged. */
int ff3d_msda_gather_rows(const float* value, const float* ref_pts, const float* off, int64_t off_ld, const float* logits,
                          int64_t logits_ld, float* out, int64_t out_ld, int groups, int B, int Nv, int Nq, int heads, int C, int L,
                          int P, const int32_t* level_hw_host, ff3d_stream_t stream);
/* ff3d_local_attention_pair (round 6, csrc/locatt_mfma.hip): the same operator (EU:158-161, k = 9) on the fp16 matrix cores with
 *   fp32-class accuracy, on the operands the neck's 1x1 GEMMs already produce: query / key / value as NHWC (hi, lo') pairs (B*H*W, C)
 *   (ZERO-ROW CONTRACT: the key planes are followed by one zero row - out-of-map window pixels read it: score 0, still part of the
 *   softmax, as kernels.cuh:30-40), q_exp / k_exp their exponents (NULL: 0); result = the context as an NHWC pair (B*H*W, C) carrying
 *   the VALUE's exponent (a convex combination of values).  `workspace`: ff3d_local_attention_pair_workspace_halfs(B, C, H, W) fp16
 *   values (the value pair rewritten pixel-contiguous per channel with a zero border).  C % 32 == 0. */
int64_t ff3d_local_attention_pair_workspace_halfs(int B, int C, int H, int W);
int ff3d_local_attention_pair(const void* q_hi, const void* q_lo, const int32_t* q_exp, const void* k_hi, const void* k_lo,
                              const int32_t* k_exp, const void* v_hi, const void* v_lo, void* workspace, void* out_hi, void* out_lo,
                              int B, int C, int H, int W, int k, float scale, ff3d_stream_t stream);
/* ff3d_locatt_ck2c_loc = kernels.cuh:82-119 ck2c_loc, the one kernel the extension's backward entry points add to the two
 * above (localAttention.cpp:17-26, 40-59; similarFunction / weightingFunction.backward, EU:72-106):
 *       y[b,c,h,w] = sum_k x[b,c,h-dy,w-dx] * weight[b,h-dy,w-dx,k]     ((dy, dx) = offset of window entry k from the centre)
 *   similar_backward(x_loc, g, is_ori=true)   = ff3d_locatt_weighting(x_loc, g)      grad of x_ori
 *   similar_backward(x_ori, g, is_ori=false)  = ff3d_locatt_ck2c_loc(x_ori, g)       grad of x_loc
 *   weighting_backward_ori(x_weight, g)       = ff3d_locatt_ck2c_loc(g, x_weight)    grad of x_ori
 *   weighting_backward_weight(x_ori, g)       = ff3d_locatt_similar(g, x_ori)        grad of x_weight */
int ff3d_locatt_ck2c_loc(const float* x, const float* weight, float* y, int B, int C, int H, int W, int kH, int kW,
                         ff3d_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * LSS pillar pooling - counterpart of the reference's CUDA extension models/utils/ops/bev_pool
 * (`bev_pool_forward`, src/bev_pool.cpp:21-53 -> bev_pool_kernel, src/bev_pool_cuda.cu:20-42).
 *   x (n, c) point features sorted by cell rank; geom_feats (n, 4) int32 (x, y, z, b) cell of every point;
 *   interval_starts / interval_lengths (n_intervals) int32; out (b, d, h, w, c) fp32, ZERO-INITIALISED by the
 *   caller (as torch::zeros in the reference glue): out[b, z, x, y, :] = sum of the interval's rows.  c % 4 == 0. */
int ff3d_bev_pool(const float* x, const int32_t* geom_feats, const int32_t* interval_starts,
                  const int32_t* interval_lengths, float* out, int b, int d, int h, int w, int n, int c,
                  int n_intervals, ff3d_stream_t stream);
/* ff3d_bev_pool_bwd = bev_pool_ext.bev_pool_backward (bev_pool.cpp:55-88, bev_pool_cuda.cu:61-84 bev_pool_grad_kernel; reached
 * from QuickCumsumCuda.backward, bev_pool_op.py:72-88): x_grad (n, c) row i = the out_grad (b, d, h, w, c) row of the cell of
 * the interval that contains point i.  Every element of x_grad is written exactly once. */
int ff3d_bev_pool_bwd(const float* out_grad, const int32_t* geom_feats, const int32_t* interval_starts,
                      const int32_t* interval_lengths, float* x_grad, int b, int d, int h, int w, int n, int c,
                      int n_intervals, ff3d_stream_t stream);

/* Lift-Splat-Shoot camera branch (necks/lss.py), two entry points.
 *
 * ff3d_lss_cells - frustum geometry + voxel binning (get_geometry lss.py:232-276 and the binning / range filter of
 * voxel_pooling lss.py:324-337) in one pass: for every frustum point e = (((b*N + n)*fH + h)*fW + w)*D + d
 *     p = (xs[w], ys[h], ds[d]);  p = post_rots_inv @ (p - post_trans)   (image augmentation, either may be NULL)
 *     p = (p.x*p.z, p.y*p.z, p.z);  g = rots @ p + trans;  g = extra_rots @ g + extra_trans   (either may be NULL)
 *     c = trunc((g - lower) / dx);  keys[e] = ((b*nz + c.z)*nx + c.x)*ny + c.y  or  B*nz*nx*ny when c is outside the grid
 *   rots / post_rots_inv / extra_rots (B*N, 3, 3), trans / post_trans / extra_trans (B*N, 3) device fp32;
 *   xs (fW), ys (fH), ds (D): the axes of the module's `frustum` parameter (lss.py:217-230);
 *   lower_host = bx - dx/2, dx_host, nx_host (x, y, z): HOST arrays of 3 (gen_dx_bx, lss.py:82-87);  keys (B*N*fH*fW*D) int32.
 *
 * ff3d_lss_splat - fused lift + splat (lss.py:132-141 outer product + :339-362 pooling / the bev_pool extension) without
 * materialising the per-point features:
 *     out[cell, :] = sum over the (pixel, depth-bin) entries of the cell of depth[pixel, d] * feat[pixel, :]
 *   feat   rows of C floats, row p at feat + p*feat_ld (e.g. the feature columns of the depthnet GEMM output)
 *   depth  (P, D) softmax-ed depth distribution per pixel
 *   src    int32 entry ids (pixel*D + d) sorted by cell key;  cell_offsets (n_cells + 1) int32: cell c owns
 *          src[cell_offsets[c] .. cell_offsets[c+1])
 *   out    (n_cells, C), every row written (zeros for empty cells).   C % 4 == 0, C <= 256, feat_ld % 4 == 0. */
int ff3d_lss_cells(const float* rots, const float* trans, const float* post_rots_inv, const float* post_trans,
                   const float* extra_rots, const float* extra_trans, const float* xs, const float* ys, const float* ds,
                   int B, int N, int D, int fH, int fW, const float* lower_host, const float* dx_host,
                   const int32_t* nx_host, int32_t* keys, ff3d_stream_t stream);
int ff3d_lss_splat(const float* feat, int64_t feat_ld, const float* depth, int D, const int32_t* src,
                   const int32_t* cell_offsets, float* out, int C, int n_cells, ff3d_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * fp32-class dense layers on the fp16 matrix cores (3-pass hi/lo split, fp32 accumulation; error ~2^-22 relative per
 * product).  Counterparts of the framework fp32 conv / linear calls behind the head's ConvModule / Linear layers
 * (heatmap head FD:202-229, BEV pyramid FD:150-162, value_proj / roi_mlp of the decoder).
 *
 * RANGE NORMALISATION.  fp16 spans 2^-14 .. 65504, the reference's fp32 arithmetic 2^-126 .. 3e38.  Every split operand
 *   therefore carries one power-of-two exponent e (an int32 scalar in DEVICE memory):
 *       x = 2^e * (hi + lo' / 2048),   |x| * 2^-e < 2^15,
 *   chosen from the tensor's magnitude so that the scaled values sit at the top of fp16's range.  Scaling by a power of
 *   two is exact, so results are independent of e wherever no value falls into fp16's subnormals, and those are below
 *   2^-29 of the tensor's maximum: the arithmetic stays fp32-class for inputs of ANY magnitude (tests: inputs x 1e5 and
 *   x 1e-6).  Nothing is read back to the host:
 *     - weights: e (and a bound for the layer output, below) computed once per weight load on the device;
 *     - fp32 -> pair conversion of an external tensor (ff3d_split_f16): the conversion runs with a GUESSED exponent
 *       (`hint`, persistent per call site; last call's value) while it measures max|x|; a one-thread kernel then checks the
 *       guess (2^5 <= max|x| * 2^-e < 2^15) and, only if it fails, a second conversion pass runs with the right exponent
 *       (the pass is always launched and exits at once otherwise - no host decision, graph-capturable).  Steady state:
 *       one pass, as without the guard;
 *     - layer outputs written as pairs: e_out from the guaranteed bound |out| <= 2^(e_in+15) * L1(W) + max|bias|
 *       (+ 2^(e_res+15)), L1(W) = the largest row sum of |W|, computed in the kernel from device scalars; the bound is loose
 *       by ~2^5 for a 2304-term dot product, i.e. costs 5 of fp16's 29 binades of head-room, nothing in precision.
 *   ff3d_scale_t collects the device scalars of one launch; a NULL ff3d_scale_t* (or NULL members) means exponent 0 /
 *   "do not write". */
typedef struct {
  const int32_t* a_exp;   /* exponent of the activation / A operand pair                                   (NULL: 0) */
  const int32_t* a2_exp;  /* ff3d_dwconv3x3_pair: exponent of the second concatenated input               (NULL: 0) */
  const int32_t* w_exp;   /* exponent of the weight pair                                                   (NULL: 0) */
  const float* w_bound;   /* 2 floats {L1(W), max|bias|} (real units): needed when out_exp != NULL                   */
  const int32_t* res_exp; /* exponent of the residual pair                                                 (NULL: 0) */
  int32_t* out_exp;       /* written: exponent of the pair output; for fp32 outputs the bound exponent, |out| < 2^(e+15)
                             (NULL: pair outputs are written unscaled, e = 0)                                         */
} ff3d_scale_t;

/* ff3d_split_f16: x fp32 -> hi = fp16(x * 2^-e), lo' = fp16((x * 2^-e - hi) * 2048).  to_nhwc = 1: x is (B, C, HW) NCHW
 *   and hi / lo are written as (B, HW, C); to_nhwc = 0: plain element order (B*C*HW elements, multiple of 4).  (The zero
 *   row of the contract below is the caller's: allocate one row more and clear it.)
 *   hint     FF3D_SPLIT_HINT_INTS int32 in device memory, persistent per call site, zero-initialised:
 *            {guessed e, max|x| bits of the last call, redo flag, -} followed by 64 maximum slots 256 bytes apart
 *   out_exp  1 int32: the exponent the planes were finally written with.   hint == NULL: e = 0, no guard (out_exp unused).
 * ff3d_conv3x3_f16x3: 3x3 convolution, padding 1, stride 1 or 2, on split NHWC activations (B, H, W, C) and split
 *   weights (N, 3, 3, C) [= (N, 9*C) with the filter tap major]; out (B, N, Ho, Wo) fp32 NCHW = conv + bias[n],
 *   optionally ReLU.  C % 32 == 0.
 * ff3d_gemm_f16x3: out (M, N) fp32 = A (M, K) @ W (N, K)^T + bias, optionally ReLU; K % 32 == 0.  ksplit in 1..64: for
 *   long-K GEMMs whose tile count does not fill the chip (roi_mlp.0 at small batch: 20 tiles, 1176 K-steps each) the K
 *   range is cut into `ksplit` slices computed by separate blocks; each slice writes its partial (M, N) plane into
 *   `workspace` (ksplit*M*N floats, caller-owned) and a second kernel adds the planes IN SLICE ORDER (deterministic) with
 *   bias / ReLU fused.  ksplit = 1: workspace unused (may be NULL).
 * ZERO-ROW CONTRACT of both: every operand plane is followed in memory by one row of zeros that the caller provides -
 *   activations (B*H*W + 1, C), conv weights (N + 1, 9*C), GEMM operands (M + 1, K) / (N + 1, K) - the kernel reads it
 *   for the convolution padding and for ragged M / N tiles (addresses are plane base + 32-bit byte offset, so a plane
 *   including its zero row must stay below 4 GiB). */
int ff3d_split_f16(const float* x, void* hi, void* lo, int B, int C, int HW, int to_nhwc, int32_t* hint, int32_t* out_exp,
                   ff3d_stream_t stream);
/* ff3d_split_f16_nhwc_group: ff3d_split_f16(to_nhwc = 1) for n (1 .. 4) maps of ONE shape (B, C, HW) in one launch per pass -
 *   HOST arrays of n pointers; every member has its own planes, hint and out_exp (both required).  The stage maps a multi-stage
 *   head receives (FD:522-528: pts_feat_conv + the per-stage maps + the extra map): 3 launches instead of 3 n at small batches. */
int ff3d_split_f16_nhwc_group(int n, const float* const* x, void* const* hi, void* const* lo, int B, int C, int HW,
                              int32_t* const* hint, int32_t* const* out_exp, ff3d_stream_t stream);
int ff3d_conv3x3_f16x3(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                       int apply_relu, float* out, int B, int C, int H, int W, int N, int stride,
                       const ff3d_scale_t* scale_host, ff3d_stream_t stream);
int ff3d_gemm_f16x3(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias,
                    int apply_relu, float* out, int M, int N, int K, int ksplit, float* workspace,
                    const ff3d_scale_t* scale_host, ff3d_stream_t stream);
/* ff3d_conv3x3_f16x3_splitk (ABI 2.10): ff3d_conv3x3_f16x3 with the K = 9*C walk cut into `ksplit` (2 .. 64, <= 9*C/32) slices
 *   computed by separate blocks - for convs whose output is only a few 128 x 128 tiles (the BEV pyramid's stride-2 convs of
 *   focal_decoder.py:150-162 at 1 - 4 frames: 90 x 90 / 45 x 45 output pixels).  Each slice writes its raw partial sums to plane s
 *   of `workspace` (ksplit * B*Ho*Wo * N floats, caller-owned, 16-byte aligned); a second kernel adds the planes IN SLICE ORDER
 *   (deterministic), applies scale / bias / ReLU and writes out (B, N, Ho, Wo) fp32 NCHW.  Same operands, zero-row contract and
 *   scale block as ff3d_conv3x3_f16x3; the fp32 sum over K is grouped by slice, so results agree with the one-pass kernel to
 *   fp32 rounding, not bit for bit. */
int ff3d_conv3x3_f16x3_splitk(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                              int apply_relu, float* out, int B, int C, int H, int W, int N, int stride, int ksplit,
                              float* workspace, const ff3d_scale_t* scale_host, ff3d_stream_t stream);
/* ff3d_gemm_f16x3_rowbias: out (nbatch*rows, N) fp32 = A (nbatch*rows, K) @ W (N, K)^T + bias_tab[row within the frame, n]
 *   with bias_tab (rows, N) fp32 shared by the nbatch frames.  The value projections of mmcv MultiScaleDeformableAttention
 *   for ALL decoder stages and layers in one launch: `value_proj(feats + bev_pos_embed)` (FD:886, FD:927-933) is linear, so
 *   the positional term and the bias move into the weight-only table pos_embed @ W^T + b and the GEMM runs over the raw
 *   pyramid once.  Tiles never straddle frames and are walked frame-fastest (table tile reused from L2). */
int ff3d_gemm_f16x3_rowbias(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias_tab,
                            float* out, int nbatch, int rows, int N, int K, const ff3d_scale_t* scale_host,
                            ff3d_stream_t stream);
/* ff3d_gemm_f16x3_fused: ff3d_gemm_f16x3 with the epilogue a 1x1-conv layer of an NHWC pair pipeline needs: act = 0 none,
 *   1 ReLU, 2 ReLU6; optional residual pair (M, N) added before the activation; result as fp32 (M, N) in `out`, or as the
 *   (hi, lo') pair in (`out_hi`, `out_lo`) (N even; exactly one of the two forms; ksplit = 1 with a pair output or a
 *   residual).  1x1 convolutions of torchvision `mobilenetv2.InvertedResidual` inside FocalEncoderLayer
 *   (focal_encoder.py:33-36) with BatchNorm folded. */
int ff3d_gemm_f16x3_fused(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias,
                          int act, const void* res_hi, const void* res_lo, float* out, void* out_hi, void* out_lo, int M,
                          int N, int K, int ksplit, float* workspace, const ff3d_scale_t* scale_host, ff3d_stream_t stream);
/* ff3d_linear_f16x3: out (M, N) fp32 = act(A (M, K) fp32 @ W (N, K)^T + bias), act = 0 none | 1 ReLU; rows of A / out at
 *   strides lda / ldc floats (column blocks of wider tensors are fine).  The query-side projections of the decoder - mmcv
 *   `MultiheadAttention` in/out projections, `MultiScaleDeformableAttention.sampling_offsets / attention_weights /
 *   output_proj`, `FFN`, the positional MLPs (UT:16-28), roi_mlp.1-2 (FD:186-200) and the prediction heads' first layer
 *   (DU:510-539); reached from FD:870-871, 914-922, 927-933, 939 - which the reference stack runs as fp32 hipBLASLt GEMMs.
 *   Split-fp16 arithmetic (three fp16 MFMA passes, fp32 accumulate, fp32-class accuracy) with the ACTIVATION operand taken
 *   as plain fp32: the kernel normalises every ROW by its own power of two (any fp32 magnitude, exact scaling) and splits it
 *   while staging it; weights are the (hi, lo') planes of an (N + 1, K) split (row N all zero) with exponent *w_exp (NULL: 0).
 *   K % 32 == 0; a, out, bias 16-byte aligned, lda, ldc % 4 == 0. */
int ff3d_linear_f16x3(const float* a, int64_t lda, const void* w_hi, const void* w_lo, const int32_t* w_exp,
                      const float* bias, int act, float* out, int64_t ldc, int M, int N, int K, ff3d_stream_t stream);
/* ff3d_linear_dual_f16x3: ff3d_linear_f16x3 whose output columns n >= n_split (a multiple of 128, 0 < n_split < N) are
 *   computed from a SECOND activation a2 (same shape and lda): out[:, :n_split] = a W[:n_split]^T, out[:, n_split:] = a2
 *   W[n_split:]^T.  The in-projection of torch `nn.MultiheadAttention` as mmcv `MultiheadAttention` calls it (query = key =
 *   x + query_pos, value = x: Appendix A.2; FD:870-871 through BaseTransformerLayer) - q | k | v in one launch. */
int ff3d_linear_dual_f16x3(const float* a, const float* a2, int n_split, int64_t lda, const void* w_hi, const void* w_lo,
                           const int32_t* w_exp, const float* bias, int act, float* out, int64_t ldc, int M, int N, int K,
                           ff3d_stream_t stream);
/* ff3d_linear_kslices_f16x3 (ABI 2.11): the K walk of a long-K projection of FEW rows cut into `kslices` (2 .. 16) slices that run as
 *   separate column blocks: out[m, s * N + n] = sum_{k < K} a[m, s * K + k] * W'[s * N + n, k], where the planes hold W' = the K slices
 *   of the layer's (N, kslices * K) weight stacked along the rows ((kslices * N + 1, K) with the zero row; ONE exponent).  N % 128 == 0,
 *   K % 32 == 0, lda >= kslices * K, ldc >= kslices * N; no bias / activation.  fc2 of mmcv `FFN` (embed 256, hidden 1024; Appendix A.2)
 *   at 1 - 4 frames, followed by ff3d_sum_add_layer_norm: each of the 38 row-owning blocks of ff3d_linear_add_ln_f16x3 streams the
 *   whole 1 MB weight there (23 us at 600 rows); sliced, four times as many blocks stream a quarter each. */
int ff3d_linear_kslices_f16x3(const float* a, int64_t lda, int kslices, const void* w_hi, const void* w_lo, const int32_t* w_exp,
                              float* out, int64_t ldc, int M, int N, int K, ff3d_stream_t stream);
/* ff3d_linear_add_ln_f16x3: out (M, N) = LayerNorm(residual + a W^T + bias) * gamma + beta over the N = 256 columns (eps inside
 *   the square root, biased variance: torch `nn.LayerNorm`), and with out_pos also out_pos = out + pos.  One decoder-layer step
 *   of mmcv `BaseTransformerLayer` in post-norm order ('self_attn' | 'cross_attn' | 'ffn' followed by 'norm': the attention's
 *   output projection / the FFN's second layer, the identity add and the norm; `+ query_pos` is the next operation's first
 *   line) in one launch.  residual, pos, out, out_pos contiguous (M, N); every pointer 16-byte aligned. */
int ff3d_linear_add_ln_f16x3(const float* a, int64_t lda, const void* w_hi, const void* w_lo, const int32_t* w_exp,
                             const float* bias, const float* residual, const float* gamma, const float* beta, float eps,
                             const float* pos, float* out, float* out_pos, int M, int N, int K, ff3d_stream_t stream);
/* ff3d_linear_rows (round 5): the same projections as the three entry points above on a ROW-OWNING tiling (a 512-thread block owns
 *   16 * MT rows x 256 columns, MT chosen so that the grid is one full round of the chip where possible) and in two arithmetics:
 *     w_lo != NULL  split-fp16 (fp32-class) exactly as ff3d_linear_f16x3: (w_hi, w_lo, *w_exp) = the (N + 1, K) split planes;
 *     w_lo == NULL  bf16 operands on v_mfma_f32_16x16x32_bf16 (BASELINE.json configs[4], "bf16 QKV/FFN on MFMA"; the reference has
 *                   no reduced-precision mode - FD:186-200, 304, 927-933 are the call sites): w_hi = the (N + 1, K) bf16 plane
 *                   (row N all zero), the activation is rounded to bf16 (round-to-nearest-even) while it is staged, products
 *                   exact, fp32 accumulation, bias added in fp32, ONE rounding of the result to bf16, ReLU on the rounded
 *                   value; the result is stored as fp32 (oracle/ff3d_oracle.py lin(lowp=True)).
 *   out (M, N) fp32 = act(a W^T + bias), rows of a / out at strides lda / ldc floats; with a2 (n_split a multiple of 256, 0 <
 *   n_split < N) the columns n >= n_split are computed from a2 (q | k | v of nn.MultiheadAttention in one launch).
 *   residual != NULL: the LayerNorm form of ff3d_linear_add_ln_f16x3 (N = 256, ldc = N, act = 0, no a2): out = LayerNorm(residual +
 *   result) * gamma + beta, and with out_pos also out_pos = out + pos - in the bf16 arithmetic the rounding to bf16 precedes the
 *   residual add.  K % 32 == 0; pointers 16-byte aligned, lda, ldc % 4 == 0. */
int ff3d_linear_rows(const float* a, const float* a2, int n_split, int64_t lda, const void* w_hi, const void* w_lo,
                     const int32_t* w_exp, const float* bias, int act, const float* residual, const float* gamma,
                     const float* beta, float eps, const float* pos, float* out, float* out_pos, int64_t ldc, int M, int N,
                     int K, ff3d_stream_t stream);
/* ff3d_ffn_rows (round 5): the feed-forward step of a post-norm decoder layer in ONE launch (mmcv `FFN` with two fcs + the identity
 *   add + the 'norm' that follows it in `BaseTransformerLayer`'s operation order; the reference reaches it through FD:927-933):
 *     out (M, 256) = LayerNorm(residual + relu(x W1^T + b1) W2^T + b2) * gamma + beta,     out_pos = out + pos (optional),
 *   split-fp16 arithmetic (fp32-class) as ff3d_linear_f16x3 with ONE difference: the low parts are UNSCALED (x = 2^e (hi + lo), lo =
 *   fp16(x 2^-e - hi); range normalisation puts the largest |x 2^-e| of a row in [2^13, 2^14), so lo is a normal fp16 number for
 *   every element within 2^-16 of the row maximum) and the three MFMA passes add into one fp32 accumulator.  x is normalised per row
 *   (one power of two over its 256 columns), the `hidden`-wide activation per row and 128-unit chunk; it never leaves the CU.
 *   x (M, 256) fp32 rows at stride lda; W1 (hidden, 256) and W2 (256, hidden) arrive as K-STEP-TILED split planes: plane[ks][n][32]
 *   fp16 = W[n][32 ks .. 32 ks + 31] scaled by 2^-*w_exp, hi and UNSCALED lo (w1t_*: 8 x hidden x 32, w2t_*: hidden / 32 x 256 x 32;
 *   no zero row - every tile is complete), so that a wave's fragment load is one contiguous KiB and the weights need no LDS.  hidden % 128 == 0; residual, pos, out, out_pos contiguous
 *   (M, 256); b2 may be NULL; pointers 16-byte aligned, lda % 4 == 0. */
int ff3d_ffn_rows(const float* x, int64_t lda, const void* w1t_hi, const void* w1t_lo, const int32_t* w1_exp, const float* b1,
                  int hidden, const void* w2t_hi, const void* w2t_lo, const int32_t* w2_exp, const float* b2, const float* residual,
                  const float* gamma, const float* beta, float eps, const float* pos, float* out, float* out_pos, int M,
                  ff3d_stream_t stream);
/* ff3d_gemm_bf16 (round 5): out (M, N) = act(A (M, K) @ W (N, K)^T + bias) on v_mfma_f32_16x16x32_bf16 with BOTH operands given as
 *   bf16 planes in device memory (each followed by one zero row, ZERO-ROW CONTRACT): exact products, fp32 accumulation, bias added
 *   in fp32, ONE rounding of the result to bf16, ReLU on the rounded value (oracle/ff3d_oracle.py lin(lowp=True); BASELINE.json
 *   configs[4] "bf16 QKV/FFN on MFMA" - the reference has no reduced-precision mode, its call sites are mmcv MSDA `value_proj`
 *   (FD:886, 927-933) and roi_mlp.0 (FD:186-200, 914-922), fp32 hipBLASLt GEMMs there).  Exactly one result form: `out` fp32 rows
 *   holding the bf16 values, or `out_bf16` bf16 rows (N % 4 == 0; what ff3d_msda_fused_fwd reads with value_dtype FF3D_BF16).
 *   K = 128 / 256 with M >= 32 768 runs weight-stationary (weights in registers, A streamed once per 256 columns); long K with few
 *   tiles takes `ksplit` K-slices through `workspace` (ksplit, M, N) fp32 (fp32 result form only).  K % 32 == 0. */
int ff3d_gemm_bf16(const void* a, const void* w, const float* bias, int apply_relu, float* out, void* out_bf16, int M, int N, int K,
                   int ksplit, float* workspace, ff3d_stream_t stream);
/* ff3d_dwconv3x3_pair: depthwise 3x3 conv (stride 1, padding 1) + bias + activation (0 / 1 ReLU / 2 ReLU6) over the channel
 *   concatenation of one or two NHWC pairs (B*H*W, C0) and (B*H*W, C1) (C1 = 0: single input) -> pair (B*H*W, C0 + C1);
 *   weight (C0 + C1, 9) fp32 with BatchNorm folded.  Channel counts multiples of 8.  The middle layer of InvertedResidual.
 *   scale: a_exp / a2_exp = the inputs' exponents, w_bound = {max row sum of |weight|, max|bias|}, out_exp.
 * ff3d_unsplit_f16: NHWC pair (B, HW, C) with exponent *exp (NULL: 0) -> fp32 NCHW (B, C, HW): back to the reference's
 *   tensor boundary. */
int ff3d_dwconv3x3_pair(const void* x0_hi, const void* x0_lo, int C0, const void* x1_hi, const void* x1_lo, int C1,
                        const float* weight, const float* bias, int act, void* out_hi, void* out_lo, int B, int H, int W,
                        const ff3d_scale_t* scale_host, ff3d_stream_t stream);
int ff3d_unsplit_f16(const void* hi, const void* lo, const int32_t* exp, float* out, int B, int C, int HW,
                     ff3d_stream_t stream);
/* ff3d_conv3x3_f16x3_split_out: as ff3d_conv3x3_f16x3, but the result is written as the (hi, lo') pair of NHWC planes
 *   (B*Ho*Wo [+ the caller's zero row], N) that a following split-fp16 layer consumes (N even).
 * ff3d_conv3x3_small_f16x3: the heatmap head's last layer (FD:213-220): conv3x3 stride 1 padding 1 with K <= 16 output
 *   channels on such a pair; weights (16 + 1, 9, C): rows K..15 and the trailing row zero; out (B, K, H, W) fp32. */
int ff3d_conv3x3_f16x3_split_out(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo,
                                 const float* bias, int apply_relu, void* out_hi, void* out_lo, int B, int C, int H,
                                 int W, int N, int stride, const ff3d_scale_t* scale_host, ff3d_stream_t stream);
/* ff3d_conv3x3_halo_f16x3: the stride-1 case of ff3d_conv3x3_f16x3 / _split_out in halo-tile form (each activation is
 *   staged once per 32-channel chunk instead of once per filter tap; 4 x 64 pixel x 128 channel tiles): same operands
 *   and zero-row contract; exactly one of `out` (NCHW fp32) or (`out_hi`, `out_lo`) (NHWC pair, N even) is non-NULL. */
int ff3d_conv3x3_halo_f16x3(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                            int apply_relu, float* out, void* out_hi, void* out_lo, int B, int C, int H, int W, int N,
                            const ff3d_scale_t* scale_host, ff3d_stream_t stream);
/* ff3d_conv3x3_halo_f16x3_nhwc (round 5): the same convolution with the result as NHWC fp32 (B, H, W, N) rows - the camera maps of
 *   `shared_conv_img` (necks/focal_encoder.py:143-147) in the layout the projection sampler gathers from (EU:236-247). */
int ff3d_conv3x3_halo_f16x3_nhwc(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                                 int apply_relu, float* out_nhwc, int B, int C, int H, int W, int N,
                                 const ff3d_scale_t* scale_host, ff3d_stream_t stream);
/* ff3d_conv3x3_halo_f16x3_tiled (round 5): ff3d_conv3x3_halo_f16x3 / _nhwc with the weight planes in K-STEP TILES: wt[tile][n][32]
 *   fp16 = w[n][tap][c0 .. c0 + 31] with tile = tap * (C / 32) + c0 / 32 and N + 1 rows per tile (row N all zero - the zero-row
 *   contract per tile), i.e. (9 C / 32, N + 1, 32); same exponent.  The 128 weight rows a block stages per step are contiguous.
 *   Exactly one of `out` (NCHW fp32), (`out_hi`, `out_lo`) (NHWC pair, N even) or `out_nhwc` (NHWC fp32) is non-NULL.  Results
 *   bit-identical to the row-major entry points. */
int ff3d_conv3x3_halo_f16x3_tiled(const void* x_hi, const void* x_lo, const void* wt_hi, const void* wt_lo, const float* bias,
                                  int apply_relu, float* out, void* out_hi, void* out_lo, float* out_nhwc, int B, int C, int H,
                                  int W, int N, const ff3d_scale_t* scale_host, ff3d_stream_t stream);
/* ff3d_conv3x3_halo_f16x3_tiled over the caller's NCHW fp32 activation x (B, C, H, W) - the fp32 -> pair conversion pass in front of a wide
 *   stride-1 3x3 convolution (FD:202-212: a heatmap head's first conv on a stage map) folded into the convolution: the block converts its
 *   halo on the way into LDS.  Bit-identical to ff3d_split_f16(to_nhwc = 1) + ff3d_conv3x3_halo_f16x3_tiled with the same exponent.
 *   hint   the call site's persistent exponent record, FF3D_SPLIT_HINT_INTS int32 in device memory, zero-initialised - the record of
 *          ff3d_split_f16, same protocol moved into the convolution: it runs with the guessed exponent while its requests measure max|x|,
 *          a one-wave kernel checks the guess, and a second launch of the convolution recomputes only if the check flagged it (always
 *          launched, exits at once otherwise: no host decision, graph-capturable).  hint[0] = the exponent in use.
 *   scale_host->a_exp is ignored; w_exp / w_bound / out_exp as for the pair form.  Weights K-step-tiled (wt[tile][n][32], N + 1 rows).
 *   Exactly one of out (B, N, H, W) fp32 / (out_hi, out_lo) NHWC pair is non-NULL.  C % 32 == 0, C * H * W * 4 < 2^31.
 *   FF3D_ERR_UNSUPPORTED where the pair form would run its 8 x 32 tile geometry (maps that 4 x 64 tiles pad by >= 1 % more, e.g. 468 x 468):
 *   convert and call the pair form there. */
int ff3d_conv3x3_halo_f16x3_nchwsrc(const float* x, int32_t* hint, const void* wt_hi, const void* wt_lo, const float* bias,
                                    int apply_relu, float* out, void* out_hi, void* out_lo, int B, int C, int H, int W, int N,
                                    const ff3d_scale_t* scale_host, ff3d_stream_t stream);
int ff3d_conv3x3_small_f16x3(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                             float* out, int B, int C, int H, int W, int K, const ff3d_scale_t* scale_host,
                             ff3d_stream_t stream);
/* ff3d_conv3x3_small_f16x3_tiled (round 5): ff3d_conv3x3_small_f16x3 with the class-padded weight planes in CHUNK TILES: wt[c0 / 32][tap]
 *   [class 0 .. 15][32] fp16 = w[class][tap][c0 .. c0 + 31], i.e. (C / 32, 9, 16, 32) - what a block stages per 32-channel chunk is one
 *   contiguous 9 216-byte run per plane.  Results bit-identical to the row-major entry point. */
int ff3d_conv3x3_small_f16x3_tiled(const void* x_hi, const void* x_lo, const void* wt_hi, const void* wt_lo, const float* bias,
                                   float* out, int B, int C, int H, int W, int K, const ff3d_scale_t* scale_host,
                                   ff3d_stream_t stream);
/* ff3d_conv3x3_halo_f16x3_group / ff3d_conv3x3_small_f16x3_group: n (1 .. 4) convolutions of ONE shape - own inputs, weights,
 *   outputs and scale records, passed as HOST arrays of n pointers - in one launch; argument meaning per member as in
 *   ff3d_conv3x3_halo_f16x3 / ff3d_conv3x3_small_f16x3 (halo form: either every out[g] or every (out_hi[g], out_lo[g])).  The
 *   heatmap heads of the multi-stage head (FD:587-668 evaluates `heatmap_head` / `heatmap_head_img[i]` on S different stage
 *   maps before the stage loop): at 1 - 8 frames three grids of 4.2 rounds of blocks become one of 12.7. */
int ff3d_conv3x3_halo_f16x3_group(int n, const void* const* x_hi, const void* const* x_lo, const void* const* w_hi,
                                  const void* const* w_lo, const float* const* bias, int apply_relu, float* const* out,
                                  void* const* out_hi, void* const* out_lo, int B, int C, int H, int W, int N,
                                  const ff3d_scale_t* const* scale_host, ff3d_stream_t stream);
int ff3d_conv3x3_small_f16x3_group(int n, const void* const* x_hi, const void* const* x_lo, const void* const* w_hi,
                                   const void* const* w_lo, const float* const* bias, float* const* out, int B, int C, int H,
                                   int W, int K, const ff3d_scale_t* const* scale_host, ff3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FF3D_H_ */
