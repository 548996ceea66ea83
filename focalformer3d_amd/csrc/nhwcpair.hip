// Memory-bound helpers of the NHWC (hi, lo') pair pipeline (the layout the split-fp16 MFMA kernels consume):
//   dwconv3x3_pair_kernel   depthwise 3x3 convolution (padding 1, stride 1) + folded-BatchNorm shift + ReLU / ReLU6 over one
//                           or two pair tensors taken as one channel-concatenated input (the `torch.cat` in front of the
//                           reference's MobileNetV2 blocks never materialises), pair output.  torchvision
//                           `mobilenetv2.InvertedResidual` middle layer as used by FocalEncoderLayer (focal_encoder.py:33-36).
//   unsplit_nhwc_to_nchw_kernel   (hi, lo') NHWC pair -> fp32 NCHW (the reference's tensor boundary), 64x64 transpose tiles.
// Values are reconstructed as hi + lo' / 2048 (exact to ~2^-22) and accumulated in fp32.
#include "ff3d_common.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;

struct DwParams {
  const _Float16 *x0_hi, *x0_lo, *x1_hi, *x1_lo;   // inputs (B*H*W, C0) and optionally (B*H*W, C1)
  const float *w, *bias;                           // (C0 + C1, 9) BN-folded, (C0 + C1)
  _Float16 *out_hi, *out_lo;                       // (B*H*W, C0 + C1)
  int B, H, W, C0, C1, relu;
  float upper;
};

constexpr int DW_X = 32, DW_Y = 8, DW_G = 8;       // block: 32 pixels x 8 channel groups of 8; walks 8 rows

__global__ __launch_bounds__(256) void dwconv3x3_pair_kernel(DwParams p) {
  __shared__ float s_w[DW_G * 8 * 9], s_b[DW_G * 8];
  const int C = p.C0 + p.C1, groups = C / 8;
  const int gx = blockIdx.x % ((p.W + DW_X - 1) / DW_X), gy = blockIdx.x / ((p.W + DW_X - 1) / DW_X);
  const int g0 = blockIdx.y * DW_G, b = blockIdx.z;
  const int tid = threadIdx.x, lx = tid & 31, lg = tid >> 5;
  for (int i = tid; i < DW_G * 8 * 9; i += 256) {
    const int c = g0 * 8 + i / 9;
    s_w[i] = c < C ? p.w[(long long)c * 9 + i % 9] : 0.f;
  }
  if (tid < DW_G * 8) s_b[tid] = (g0 * 8 + tid < C && p.bias) ? p.bias[g0 * 8 + tid] : 0.f;
  __syncthreads();
  const int g = g0 + lg, x = gx * DW_X + lx;
  if (g >= groups || x >= p.W) return;
  const int c = g * 8;
  // the channel group lives in input 0 or input 1
  const bool second = c >= p.C0;
  const _Float16* xh = second ? p.x1_hi : p.x0_hi;
  const _Float16* xl = second ? p.x1_lo : p.x0_lo;
  const int Cin = second ? p.C1 : p.C0, cin = second ? c - p.C0 : c;
  const float* wg = s_w + lg * 72;
  for (int ry = 0; ry < DW_Y; ++ry) {
    const int y = gy * DW_Y + ry;
    if (y >= p.H) break;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = s_b[lg * 8 + k];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= p.H) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = x + dx;
        if (xx < 0 || xx >= p.W) continue;
        const long long o = (((long long)b * p.H + yy) * p.W + xx) * Cin + cin;
        const half8 h = *reinterpret_cast<const half8*>(xh + o);
        const half8 l = *reinterpret_cast<const half8*>(xl + o);
        const int t = (dy + 1) * 3 + dx + 1;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = fmaf(wg[k * 9 + t], (float)h[k] + (float)l[k] * (1.f / 2048.f), acc[k]);
      }
    }
    half8 oh, ol;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = acc[k];
      if (p.relu) v = fminf(fmaxf(v, 0.f), p.upper);
      const _Float16 hh = (_Float16)v;
      oh[k] = hh;
      ol[k] = (_Float16)((v - (float)hh) * 2048.f);
    }
    const long long oo = (((long long)b * p.H + y) * p.W + x) * C + c;
    *reinterpret_cast<half8*>(p.out_hi + oo) = oh;
    *reinterpret_cast<half8*>(p.out_lo + oo) = ol;
  }
}

__global__ __launch_bounds__(256) void unsplit_nhwc_to_nchw_kernel(const _Float16* __restrict__ hi,
                                                                   const _Float16* __restrict__ lo,
                                                                   float* __restrict__ out, int C, int HW) {
  __shared__ float tile[64][65];                 // [pixel][channel]
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int q = ty; q < 64; q += 4) {
    const int pp = p0 + q, cc = c0 + tx;
    float v = 0.f;
    if (pp < HW && cc < C) {
      const long long o = ((long long)b * HW + pp) * C + cc;
      v = (float)hi[o] + (float)lo[o] * (1.f / 2048.f);
    }
    tile[q][tx] = v;
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    const int cc = c0 + c, pp = p0 + tx;
    if (cc < C && pp < HW) out[((long long)b * C + cc) * HW + pp] = tile[tx][c];
  }
}

}  // namespace

extern "C" int ff3d_dwconv3x3_pair(const void* x0_hi, const void* x0_lo, int C0, const void* x1_hi, const void* x1_lo,
                                   int C1, const float* weight, const float* bias, int act, void* out_hi, void* out_lo,
                                   int B, int H, int W, ff3d_stream_t stream) {
  FF3D_REQUIRE(x0_hi && x0_lo && weight && out_hi && out_lo && (C1 == 0 || (x1_hi && x1_lo)), FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && C0 > 0 && C0 % 8 == 0 && C1 >= 0 && C1 % 8 == 0 && act >= 0 && act <= 2,
               FF3D_ERR_BAD_SHAPE);
  DwParams p{static_cast<const _Float16*>(x0_hi), static_cast<const _Float16*>(x0_lo),
             static_cast<const _Float16*>(x1_hi), static_cast<const _Float16*>(x1_lo), weight, bias,
             static_cast<_Float16*>(out_hi), static_cast<_Float16*>(out_lo), B, H, W, C0, C1, act ? 1 : 0,
             act == 2 ? 6.f : INFINITY};
  const int groups = (C0 + C1) / 8;
  const dim3 grid(((W + DW_X - 1) / DW_X) * ((H + DW_Y - 1) / DW_Y), (groups + DW_G - 1) / DW_G, B);
  ff3d_clear_error();
  hipLaunchKernelGGL(dwconv3x3_pair_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return ff3d_launch_status();
}

extern "C" int ff3d_unsplit_f16(const void* hi, const void* lo, float* out, int B, int C, int HW, ff3d_stream_t stream) {
  FF3D_REQUIRE(hi && lo && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && B <= 65535 && C > 0 && HW > 0, FF3D_ERR_BAD_SHAPE);
  ff3d_clear_error();
  hipLaunchKernelGGL(unsplit_nhwc_to_nchw_kernel, dim3((HW + 63) / 64, (C + 63) / 64, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const _Float16*>(hi), static_cast<const _Float16*>(lo),
                     out, C, HW);
  return ff3d_launch_status();
}
