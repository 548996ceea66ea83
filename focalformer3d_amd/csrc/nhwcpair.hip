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
  Ff3dScale sc;                                    // a_exp / a2_exp: input exponents; w_bound; out_exp (ff3d.h)
};

constexpr int DW_L = 30;                           // pixels a thread walks along x (180 = 6 x 30)

// Thread = (channel group of 8, row strip of DW_L pixels); consecutive lanes = consecutive channel groups, so a wave reads
// whole 16*lanes-byte pixel rows (coalesced).  The thread keeps its 8 x 9 folded weights and a 3-column x 3-row window of
// reconstructed fp32 values in registers and slides it along x: 6 16-byte loads per output pixel instead of 18.
__global__ __launch_bounds__(256) void dwconv3x3_pair_kernel(DwParams p) {
  const int C = p.C0 + p.C1, groups = C / 8;
  const int strips_x = (p.W + DW_L - 1) / DW_L;
  const long long total = (long long)p.B * p.H * strips_x * groups;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int g = (int)(gid % groups);
  long long r = gid / groups;
  const int sx = (int)(r % strips_x);
  r /= strips_x;
  const int y = (int)(r % p.H), b = (int)(r / p.H);
  const int c = g * 8;
  const bool second = c >= p.C0;
  const _Float16* xh = second ? p.x1_hi : p.x0_hi;
  const _Float16* xl = second ? p.x1_lo : p.x0_lo;
  const int Cin = second ? p.C1 : p.C0, cin = second ? c - p.C0 : c;
  // range normalisation: inputs in real units = pair * 2^e_in; the output pair gets one exponent for both halves of the
  // concatenation, from the bound 2^(max(e_0, e_1) + 15) * max_c sum|w_c| + max|bias|
  const int e0 = ff3d_ld_exp(p.sc.a_exp), e1 = p.C1 ? ff3d_ld_exp(p.sc.a2_exp) : e0;
  const float sc_in = ff3d_pow2(second ? e1 : e0);
  float sc_out = 1.f;
  if (p.sc.out_exp) {
    const int e_out = ff3d_out_exp(p.sc, max(e0, e1), false, p.relu ? p.upper : INFINITY);
    sc_out = ff3d_pow2(-e_out);
    if (gid == 0) *p.sc.out_exp = e_out;
  }

  float w[8][9], bias[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    bias[k] = p.bias ? p.bias[c + k] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) w[k][t] = p.w[(long long)(c + k) * 9 + t];
  }
  const long long row_base = ((long long)b * p.H) * p.W;
  auto load_col = [&](float (&col)[3][8], int xx) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y + dy - 1;
      if (xx >= 0 && xx < p.W && yy >= 0 && yy < p.H) {
        const long long o = (row_base + (long long)yy * p.W + xx) * Cin + cin;
        const half8 h = *reinterpret_cast<const half8*>(xh + o);
        const half8 l = *reinterpret_cast<const half8*>(xl + o);
#pragma unroll
        for (int k = 0; k < 8; ++k) col[dy][k] = fmaf((float)l[k], 1.f / 2048.f, (float)h[k]) * sc_in;
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) col[dy][k] = 0.f;
      }
    }
  };
  auto emit = [&](int x, const float (&cl)[3][8], const float (&cm)[3][8], const float (&cr)[3][8]) {
    half8 oh, ol;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = bias[k];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        v = fmaf(w[k][dy * 3 + 0], cl[dy][k], v);
        v = fmaf(w[k][dy * 3 + 1], cm[dy][k], v);
        v = fmaf(w[k][dy * 3 + 2], cr[dy][k], v);
      }
      if (p.relu) v = fminf(fmaxf(v, 0.f), p.upper);
      v *= sc_out;
      const _Float16 hh = (_Float16)v;
      oh[k] = hh;
      ol[k] = (_Float16)((v - (float)hh) * 2048.f);
    }
    const long long oo = (row_base + (long long)y * p.W + x) * C + c;
    *reinterpret_cast<half8*>(p.out_hi + oo) = oh;
    *reinterpret_cast<half8*>(p.out_lo + oo) = ol;
  };
  const int x0 = sx * DW_L, x1 = min(x0 + DW_L, p.W);
  float ca[3][8], cb[3][8], cc[3][8];
  load_col(ca, x0 - 1);
  load_col(cb, x0);
  for (int x = x0; x < x1; x += 3) {              // window roles rotate: (ca, cb, cc) -> (cb, cc, ca) -> (cc, ca, cb)
    load_col(cc, x + 1);
    emit(x, ca, cb, cc);
    if (x + 1 < x1) {
      load_col(ca, x + 2);
      emit(x + 1, cb, cc, ca);
    }
    if (x + 2 < x1) {
      load_col(cb, x + 3);
      emit(x + 2, cc, ca, cb);
    }
  }
}

__global__ __launch_bounds__(256) void unsplit_nhwc_to_nchw_kernel(const _Float16* __restrict__ hi,
                                                                   const _Float16* __restrict__ lo,
                                                                   float* __restrict__ out, int C, int HW, int vec4,
                                                                   const int* __restrict__ exp) {
  __shared__ float tile[64][65];                 // [pixel][channel]
  const float sc = ff3d_pow2(ff3d_ld_exp(exp));  // real units = pair * 2^e
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
  if (vec4) {   // C % 4 == 0, HW % 4 == 0, aligned bases: 8-byte reads along channels, 16-byte writes along pixels
    const int l16 = threadIdx.x & 15, r16 = threadIdx.x >> 4;
    for (int q = r16; q < 64; q += 16) {
      const int pp = p0 + q, cc = c0 + 4 * l16;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (pp < HW && cc < C) {
        const long long o = ((long long)b * HW + pp) * C + cc;
        const uint2 h = *reinterpret_cast<const uint2*>(hi + o), l = *reinterpret_cast<const uint2*>(lo + o);
        const _Float16* hp = reinterpret_cast<const _Float16*>(&h);
        const _Float16* lp = reinterpret_cast<const _Float16*>(&l);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaf((float)lp[k], 1.f / 2048.f, (float)hp[k]) * sc;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) tile[q][4 * l16 + k] = v[k];
    }
    __syncthreads();
    for (int c = r16; c < 64; c += 16) {
      const int cc = c0 + c, pp = p0 + 4 * l16;
      if (cc < C && pp < HW)
        *reinterpret_cast<float4*>(out + ((long long)b * C + cc) * HW + pp) =
            make_float4(tile[4 * l16][c], tile[4 * l16 + 1][c], tile[4 * l16 + 2][c], tile[4 * l16 + 3][c]);
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int q = ty; q < 64; q += 4) {
    const int pp = p0 + q, cc = c0 + tx;
    float v = 0.f;
    if (pp < HW && cc < C) {
      const long long o = ((long long)b * HW + pp) * C + cc;
      v = ((float)hi[o] + (float)lo[o] * (1.f / 2048.f)) * sc;
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
                                   int B, int H, int W, const ff3d_scale_t* scale_host, ff3d_stream_t stream) {
  FF3D_REQUIRE(x0_hi && x0_lo && weight && out_hi && out_lo && (C1 == 0 || (x1_hi && x1_lo)), FF3D_ERR_NULL);
  FF3D_REQUIRE(!scale_host || !scale_host->out_exp || scale_host->w_bound, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && C0 > 0 && C0 % 8 == 0 && C1 >= 0 && C1 % 8 == 0 && act >= 0 && act <= 2,
               FF3D_ERR_BAD_SHAPE);
  DwParams p{static_cast<const _Float16*>(x0_hi), static_cast<const _Float16*>(x0_lo),
             static_cast<const _Float16*>(x1_hi), static_cast<const _Float16*>(x1_lo), weight, bias,
             static_cast<_Float16*>(out_hi), static_cast<_Float16*>(out_lo), B, H, W, C0, C1, act ? 1 : 0,
             act == 2 ? 6.f : INFINITY, ff3d_scale_from(scale_host)};
  const long long total = (long long)B * H * ((W + DW_L - 1) / DW_L) * ((C0 + C1) / 8);
  FF3D_REQUIRE((total + 255) / 256 < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  ff3d_clear_error();
  hipLaunchKernelGGL(dwconv3x3_pair_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  return ff3d_launch_status();
}

extern "C" int ff3d_unsplit_f16(const void* hi, const void* lo, const int32_t* exp, float* out, int B, int C, int HW,
                                ff3d_stream_t stream) {
  FF3D_REQUIRE(hi && lo && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && B <= 65535 && C > 0 && HW > 0, FF3D_ERR_BAD_SHAPE);
  ff3d_clear_error();
  const int vec4 = (C % 4 == 0) && (HW % 4 == 0) && ff3d_aligned16(out) && (reinterpret_cast<uintptr_t>(hi) % 8 == 0) &&
                   (reinterpret_cast<uintptr_t>(lo) % 8 == 0);
  hipLaunchKernelGGL(unsplit_nhwc_to_nchw_kernel, dim3((HW + 63) / 64, (C + 63) / 64, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const _Float16*>(hi), static_cast<const _Float16*>(lo),
                     out, C, HW, vec4, exp);
  return ff3d_launch_status();
}
