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
