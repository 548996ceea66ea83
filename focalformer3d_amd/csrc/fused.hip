// Small fused epilogues of the decoder for gfx950 (each replaces 2-4 elementwise / normalisation launches):
//   add_layer_norm : y = LayerNorm(a + b) [; y_pos = y + pos]       one wave per row, row kept in registers
//   bias_relu      : x = relu(x + bias[c]) in place on NCHW maps     16-byte vectorised
// HBM-bound elementwise work; no MFMA.
#include "ff3d_common.h"

namespace {

constexpr int LN_MAX_PER_LANE = 16;  // C <= 1024

__global__ __launch_bounds__(256) void add_layer_norm_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ pos, float* __restrict__ out,
                                                             float* __restrict__ out_pos, long long rows, int C,
                                                             float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* pa = a + row * C;
  const float* pb = b ? b + row * C : nullptr;
  float v[LN_MAX_PER_LANE];
  float sum = 0.f;
  int n = 0;
  for (int c = lane; c < C; c += 64, ++n) {
    const float x = pa[c] + (pb ? pb[c] : 0.f);
    v[n] = x;
    sum += x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)C;
  float sq = 0.f;
  for (int i = 0; i < n; ++i) {
    const float d = v[i] - mean;
    sq += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = rsqrtf(sq / (float)C + eps);
  n = 0;
  for (int c = lane; c < C; c += 64, ++n) {
    const float y = (v[n] - mean) * rstd * gamma[c] + beta[c];
    out[row * C + c] = y;
    if (out_pos) out_pos[row * C + c] = y + pos[row * C + c];
  }
}

// LayerNorm(residual + bias + sum_s parts[row, s * C + c]) - the second half of a K-sliced projection (ff3d_linear_kslices_f16x3): the
// partial columns are added in slice order (deterministic), then add_layer_norm_kernel's arithmetic.
__global__ __launch_bounds__(256) void sum_add_layer_norm_kernel(const float* __restrict__ parts, int nparts, long long ld,
                                                                 const float* __restrict__ bias, const float* __restrict__ res,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ pos, float* __restrict__ out,
                                                                 float* __restrict__ out_pos, long long rows, int C, float eps,
                                                                 int vec256) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* pa = parts + row * ld;
  if (vec256) {                                  // C = 256, 16-byte aligned operands: a lane owns 4 consecutive channels (same arithmetic)
    const int c = 4 * lane;
    float4 x = *reinterpret_cast<const float4*>(pa + c);
    for (int s = 1; s < nparts; ++s) {
      const float4 q = *reinterpret_cast<const float4*>(pa + (long long)s * 256 + c);
      x.x += q.x, x.y += q.y, x.z += q.z, x.w += q.w;
    }
    if (bias) {
      const float4 q = *reinterpret_cast<const float4*>(bias + c);
      x.x += q.x, x.y += q.y, x.z += q.z, x.w += q.w;
    }
    if (res) {
      const float4 q = *reinterpret_cast<const float4*>(res + row * 256 + c);
      x.x += q.x, x.y += q.y, x.z += q.z, x.w += q.w;
    }
    float sum = (x.x + x.y) + (x.z + x.w);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum * (1.f / 256.f);
    const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
    float sq = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = rsqrtf(sq * (1.f / 256.f) + eps);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
    const float4 y = make_float4(d0 * rstd * g.x + b.x, d1 * rstd * g.y + b.y, d2 * rstd * g.z + b.z, d3 * rstd * g.w + b.w);
    *reinterpret_cast<float4*>(out + row * 256 + c) = y;
    if (out_pos) {
      const float4 q = *reinterpret_cast<const float4*>(pos + row * 256 + c);
      *reinterpret_cast<float4*>(out_pos + row * 256 + c) = make_float4(y.x + q.x, y.y + q.y, y.z + q.z, y.w + q.w);
    }
    return;
  }
  float v[LN_MAX_PER_LANE];
  float sum = 0.f;
  int n = 0;
  for (int c = lane; c < C; c += 64, ++n) {
    float x = pa[c];
    for (int s = 1; s < nparts; ++s) x += pa[(long long)s * C + c];
    x += bias ? bias[c] : 0.f;
    x += res ? res[row * C + c] : 0.f;
    v[n] = x;
    sum += x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)C;
  float sq = 0.f;
  for (int i = 0; i < n; ++i) {
    const float d = v[i] - mean;
    sq += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = rsqrtf(sq / (float)C + eps);
  n = 0;
  for (int c = lane; c < C; c += 64, ++n) {
    const float y = (v[n] - mean) * rstd * gamma[c] + beta[c];
    out[row * C + c] = y;
    if (out_pos) out_pos[row * C + c] = y + pos[row * C + c];
  }
}

__global__ __launch_bounds__(256) void bias_relu_kernel(float* __restrict__ x, const float* __restrict__ bias,
                                                        long long n4, int HW4, int C, float upper) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i / HW4) % C);
  const float bv = bias ? bias[c] : 0.f;
  float4 v = reinterpret_cast<float4*>(x)[i];
  v.x = fminf(fmaxf(v.x + bv, 0.f), upper);
  v.y = fminf(fmaxf(v.y + bv, 0.f), upper);
  v.z = fminf(fmaxf(v.z + bv, 0.f), upper);
  v.w = fminf(fmaxf(v.w + bv, 0.f), upper);
  reinterpret_cast<float4*>(x)[i] = v;
}

__global__ __launch_bounds__(256) void bias_relu_scalar_kernel(float* __restrict__ x, const float* __restrict__ bias,
                                                               long long n, int HW, int C, float upper) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float bv = bias ? bias[(int)((i / HW) % C)] : 0.f;
  x[i] = fminf(fmaxf(x[i] + bv, 0.f), upper);
}

}  // namespace

extern "C" int ff3d_add_layer_norm(const float* a, const float* b, const float* gamma, const float* beta,
                                   const float* pos, float* out, float* out_pos, int64_t rows, int C, float eps,
                                   ff3d_stream_t stream) {
  FF3D_REQUIRE(a && gamma && beta && out && (!out_pos || pos), FF3D_ERR_NULL);
  FF3D_REQUIRE(rows > 0 && C > 0 && C <= 64 * LN_MAX_PER_LANE, FF3D_ERR_BAD_SHAPE);
  const long long blocks = (rows + 3) / 4;
  FF3D_REQUIRE(blocks < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  ff3d_clear_error();
  hipLaunchKernelGGL(add_layer_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a, b,
                     gamma, beta, pos, out, out_pos, (long long)rows, C, eps);
  return ff3d_launch_status();
}

extern "C" int ff3d_sum_add_layer_norm(const float* parts, int nparts, int64_t ld_parts, const float* bias, const float* residual,
                                       const float* gamma, const float* beta, const float* pos, float* out, float* out_pos,
                                       int64_t rows, int C, float eps, ff3d_stream_t stream) {
  FF3D_REQUIRE(parts && gamma && beta && out && (!out_pos || pos), FF3D_ERR_NULL);
  FF3D_REQUIRE(rows > 0 && C > 0 && C <= 64 * LN_MAX_PER_LANE && nparts >= 1 && nparts <= 64 && ld_parts >= (int64_t)nparts * C,
               FF3D_ERR_BAD_SHAPE);
  const long long blocks = (rows + 3) / 4;
  FF3D_REQUIRE(blocks < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  ff3d_clear_error();
  const int vec256 = C == 256 && ld_parts % 4 == 0 && ff3d_aligned16(parts) && ff3d_aligned16(gamma) && ff3d_aligned16(beta) &&
                     ff3d_aligned16(out) && (!bias || ff3d_aligned16(bias)) && (!residual || ff3d_aligned16(residual)) &&
                     (!pos || ff3d_aligned16(pos)) && (!out_pos || ff3d_aligned16(out_pos));
  hipLaunchKernelGGL(sum_add_layer_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), parts, nparts,
                     (long long)ld_parts, bias, residual, gamma, beta, pos, out, out_pos, (long long)rows, C, eps, vec256);
  return ff3d_launch_status();
}

extern "C" int ff3d_bias_relu(float* x, const float* bias, int N, int C, int HW, float upper, ff3d_stream_t stream) {
  if (!(upper > 0.f)) upper = INFINITY;   // <= 0: plain ReLU; 6 gives ReLU6
  FF3D_REQUIRE(x, FF3D_ERR_NULL);
  FF3D_REQUIRE(N > 0 && C > 0 && HW > 0, FF3D_ERR_BAD_SHAPE);
  if (HW % 4 != 0 || !ff3d_aligned16(x)) {  // odd map sizes: scalar variant
    const long long n = (long long)N * C * HW;
    const long long nb = (n + 255) / 256;
    FF3D_REQUIRE(nb < (1ll << 31), FF3D_ERR_BAD_SHAPE);
    ff3d_clear_error();
    hipLaunchKernelGGL(bias_relu_scalar_kernel, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), x, bias,
                       n, HW, C, upper);
    return ff3d_launch_status();
  }
  const long long n4 = (long long)N * C * HW / 4;
  const long long blocks = (n4 + 255) / 256;
  FF3D_REQUIRE(blocks < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  ff3d_clear_error();
  hipLaunchKernelGGL(bias_relu_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, bias, n4,
                     HW / 4, C, upper);
  return ff3d_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Box update of a decoder stage in one launch: FD:939-957 + the per-key concatenation over stages of FD:970-987.
// In: the (B, S, Nq) output of the fused prediction GEMM (channels = center 2 | height 1 | dim 3 | rot 2 | [vel 2] | heatmap K
// in the order of the config's `common_heads`), its bias, the normalised reference points and the previous stage's box.
// Out: every head's slice written at column offset q0 of its (B, n, ld) result tensor (ld = stages * Nq: no torch.cat
// afterwards), the next stage's query positions (B, Nq, 2) and query box (B, 8 | 10, Nq).  One thread per (frame, query);
// consecutive threads = consecutive queries, so every channel row is read and written coalesced.
namespace {
struct BoxUpdateParams {
  const float *raw, *bias, *ref, *prev_box;
  float *center, *height, *dim, *rot, *vel, *heat, *qpos_out, *box_out;
  int B, S, Nq, K, ld, q0, nb;
  int raw_cs, raw_qs;          // strides of raw in floats: channel, query ((B, S, Nq): Nq, 1; (B * Nq, S) rows: 1, S)
  int c_center, c_height, c_dim, c_rot, c_vel, c_heat;
  int roi_based_reg;
  float w, h;
};

__global__ __launch_bounds__(256) void box_update_kernel(BoxUpdateParams p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.B * p.Nq) return;
  const int b = i / p.Nq, q = i - b * p.Nq;
  const float* r = p.raw + (long long)b * p.S * p.Nq + (long long)q * p.raw_qs;
  auto val = [&](int c) { return r[(long long)c * p.raw_cs] + p.bias[c]; };
  auto dst = [&](float* base, int n, int c) -> float& { return base[((long long)b * n + c) * p.ld + p.q0 + q]; };
  float box[10];
  // FD:936 + 945-947: centre = prediction + (reference point * (W, H)), and it is the next stage's query position
  const float cx = val(p.c_center) + p.ref[(long long)i * 2] * p.w, cy = val(p.c_center + 1) + p.ref[(long long)i * 2 + 1] * p.h;
  dst(p.center, 2, 0) = cx, dst(p.center, 2, 1) = cy;
  p.qpos_out[(long long)i * 2] = cx, p.qpos_out[(long long)i * 2 + 1] = cy;
  box[0] = cx, box[1] = cy;
  box[2] = val(p.c_height);
  dst(p.height, 1, 0) = box[2];
  const float* pb = p.prev_box ? p.prev_box + (long long)b * p.nb * p.Nq + q : nullptr;
  const bool add = p.roi_based_reg && pb;                  // FD:949-951: dim[:2] and rot are residuals of the previous box
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    box[3 + c] = val(p.c_dim + c) + ((add && c < 2) ? pb[(long long)(3 + c) * p.Nq] : 0.f);
    dst(p.dim, 3, c) = box[3 + c];
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    box[6 + c] = val(p.c_rot + c) + (add ? pb[(long long)(6 + c) * p.Nq] : 0.f);
    dst(p.rot, 2, c) = box[6 + c];
  }
  if (p.c_vel >= 0) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      box[8 + c] = val(p.c_vel + c);
      dst(p.vel, 2, c) = box[8 + c];
    }
  }
  for (int c = 0; c < p.K; ++c) dst(p.heat, p.K, c) = val(p.c_heat + c);
  float* ob = p.box_out + (long long)b * p.nb * p.Nq + q;
  for (int c = 0; c < p.nb; ++c) ob[(long long)c * p.Nq] = box[c];
}
}  // namespace

static int box_update_launch(const float* raw, int rows_layout, const float* bias, const float* ref, const float* prev_box,
                             float* center, float* height, float* dim, float* rot, float* vel, float* heat, float* qpos_out,
                             float* box_out, int B, int S, int Nq, int K, int64_t ld, int q0, const int32_t* channel_offsets_host,
                             int roi_based_reg, float W, float H, ff3d_stream_t stream) {
  FF3D_REQUIRE(raw && bias && ref && center && height && dim && rot && heat && qpos_out && box_out && channel_offsets_host,
               FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && S > 0 && Nq > 0 && K > 0 && q0 >= 0 && q0 + Nq <= ld && ld < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  const int32_t* o = channel_offsets_host;                  // center, height, dim, rot, vel (-1: none), heatmap
  FF3D_REQUIRE((o[4] >= 0) == (vel != nullptr), FF3D_ERR_NULL);
  for (int k = 0; k < 6; ++k) FF3D_REQUIRE(o[k] < S && (o[k] >= 0 || k == 4), FF3D_ERR_BAD_SHAPE);
  BoxUpdateParams p{raw, bias, ref, prev_box, center, height, dim, rot, vel, heat, qpos_out, box_out, B, S, Nq, K, (int)ld, q0,
                    vel ? 10 : 8, rows_layout ? 1 : Nq, rows_layout ? S : 1, o[0], o[1], o[2], o[3], o[4], o[5],
                    roi_based_reg ? 1 : 0, W, H};
  ff3d_clear_error();
  hipLaunchKernelGGL(box_update_kernel, dim3((B * Nq + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return ff3d_launch_status();
}

extern "C" int ff3d_box_update(const float* raw, const float* bias, const float* ref, const float* prev_box, float* center,
                               float* height, float* dim, float* rot, float* vel, float* heat, float* qpos_out,
                               float* box_out, int B, int S, int Nq, int K, int64_t ld, int q0,
                               const int32_t* channel_offsets_host, int roi_based_reg, float W, float H,
                               ff3d_stream_t stream) {
  return box_update_launch(raw, 0, bias, ref, prev_box, center, height, dim, rot, vel, heat, qpos_out, box_out, B, S, Nq, K, ld, q0,
                           channel_offsets_host, roi_based_reg, W, H, stream);
}

// The same with raw as the (B * Nq, S) ROW-MAJOR output of a query-major GEMM (round 5: the prediction heads' second layer on
// ff3d_linear_f16x3 instead of a vendor batched GEMM that writes (B, S, Nq)).
extern "C" int ff3d_box_update_rows(const float* raw, const float* bias, const float* ref, const float* prev_box, float* center,
                                    float* height, float* dim, float* rot, float* vel, float* heat, float* qpos_out,
                                    float* box_out, int B, int S, int Nq, int K, int64_t ld, int q0,
                                    const int32_t* channel_offsets_host, int roi_based_reg, float W, float H,
                                    ff3d_stream_t stream) {
  return box_update_launch(raw, 1, bias, ref, prev_box, center, height, dim, rot, vel, heat, qpos_out, box_out, B, S, Nq, K, ld, q0,
                           channel_offsets_host, roi_based_reg, W, H, stream);
}
