// Final layer of the heatmap head on the fp16 matrix cores: out = conv3x3(y, w) + bias with K <= 16 output channels
// (10 nuScenes / 3 Waymo classes) where y - the BN + ReLU output of the head's first conv - arrives as the (hi, lo')
// fp16 NHWC pair written by ff3d_conv3x3_f16x3_split_out.  Same arithmetic as splitmm.hip (three MFMA passes
// hi*hi + (hi*lo' + lo'*hi) * 2^-11, fp32 accumulation), different data flow: with only 16 output columns an implicit GEMM
// would stream every activation nine times through the LDS DMA (9.6 GB per call), so a block owns an 8 x 32 pixel tile,
// DMAs its 10 x 34 halo ONCE per 32-channel chunk (43.5 KB for both planes, lane-linear LDS image with the XOR chunk
// swizzle of splitmm.hip applied on the source address; pixels outside the map read the plane's trailing zero row) plus
// the chunk's weights for all nine taps (18 KB), and serves the nine taps from LDS: A = 16 consecutive pixels of a
// row, B = 16 (padded) classes, K = 32 channels per v_mfma_f32_16x16x32_f16.  Replaces relu_conv3x3_small_kernel
// (fp32 MFMA, 1.25 ms at B=32) on the default path.  Reference: heatmap_head.1, FD:213-220.
#include "ff3d_common.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int TL_Y = 8, TL_X = 32, TL_HX = TL_X + 2, TL_HALO = (TL_Y + 2) * TL_HX;   // 340 halo pixels
constexpr int TL_BK = 32, TL_ACT = TL_HALO * TL_BK, TL_WT = 9 * 16 * TL_BK;          // halves per plane
constexpr int TL_ASLOTS = TL_HALO * 4, TL_WSLOTS = 9 * 16 * 4;                      // 16-byte DMA slots per plane

struct TailParams {
  const _Float16 *x_hi, *x_lo, *w_hi, *w_lo;   // x: (B*H*W + 1, C) NHWC + zero row; w: (16 + 1, 9, C) class-padded
  const float* bias;
  float* out;                                  // (B, K, H, W) fp32
  int B, C, H, W, K;
  unsigned x_zero;                             // byte offset of the activations' zero row
  Ff3dScale sc;                                // operand exponents (ff3d.h RANGE NORMALISATION)
  int w_tiled = 0;                             // round 5: weights as (C / 32, 9, 16, 32) chunk tiles - the 9 216 bytes a block stages per
                                               // chunk and plane are contiguous and already in LDS row order (ff3d_conv3x3_small_f16x3_tiled)
};

__device__ __forceinline__ int tl_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

__device__ __forceinline__ void tl_glds16(const _Float16* base, unsigned byte_off, _Float16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(base) + byte_off,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void tl_body(const TailParams& p, unsigned bid, unsigned nblk) {
  __shared__ _Float16 s_act[2][TL_ACT];        // [plane][halo pixel][32 channels]   2 x 21 760 B
  __shared__ _Float16 s_w[2][TL_WT];           // [plane][tap][class][32 channels]   2 x  9 216 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int tiles_x = (p.W + TL_X - 1) / TL_X, tiles_y = (p.H + TL_Y - 1) / TL_Y;
  const unsigned lid = ff3d_xcd_remap(bid, nblk);
  const int b = (int)(lid / (tiles_x * tiles_y)), t = (int)(lid % (tiles_x * tiles_y));
  const int ty0 = (t / tiles_x) * TL_Y, tx0 = (t % tiles_x) * TL_X;

  // DMA slot geometry (chunk-invariant): slot s = it*256 + tid -> row s >> 2, LDS chunk s & 3 <- source chunk (s & 3) ^ swz
  unsigned a_off[6], w_off[3];
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int s = it * 256 + tid, px = s >> 2, ly = px / TL_HX, lx = px - ly * TL_HX;
    const int gy = ty0 + ly - 1, gx = tx0 + lx - 1;
    const unsigned chunk_b = (unsigned)(((s & 3) ^ tl_swz(px)) * 16);
    const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    a_off[it] = (in ? (unsigned)(((b * p.H + gy) * p.W + gx) * p.C) * 2u : p.x_zero) + chunk_b;
  }
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int s = it * 256 + tid, row = s >> 2, tap = row >> 4, cls = row & 15;
    w_off[it] = (unsigned)((cls * 9 + tap) * p.C) * 2u + (unsigned)(((s & 3) ^ tl_swz(row)) * 16);
    if (p.w_tiled) w_off[it] = (unsigned)row * 64u + (unsigned)(((s & 3) ^ tl_swz(row)) * 16);
  }

  f32x4 acc_m[4], acc_x[4];                    // 4 M-tiles per wave: rows 2*wave, 2*wave + 1, x halves 0 / 16
#pragma unroll
  for (int m = 0; m < 4; ++m) acc_m[m] = f32x4{0.f, 0.f, 0.f, 0.f}, acc_x[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c0 = 0; c0 < p.C; c0 += TL_BK) {
    __syncthreads();                           // every wave finished reading the previous chunk
    const unsigned cb = (unsigned)c0 * 2u;
#pragma unroll
    for (int it = 0; it < 6; ++it)
      if (it * 256 + tid < TL_ASLOTS) {
        _Float16* dst = &s_act[0][0] + (it * 256 + wave * 64) * 8;     // wave-uniform; the DMA adds lane * 16 B
        tl_glds16(p.x_hi, a_off[it] + cb, dst);
        tl_glds16(p.x_lo, a_off[it] + cb, dst + TL_ACT);
      }
#pragma unroll
    for (int it = 0; it < 3; ++it)
      if (it * 256 + tid < TL_WSLOTS) {
        _Float16* dst = &s_w[0][0] + (it * 256 + wave * 64) * 8;
        const unsigned wb = p.w_tiled ? (unsigned)(c0 >> 5) * (unsigned)(TL_WT * 2) : cb;
        tl_glds16(p.w_hi, w_off[it] + wb, dst);
        tl_glds16(p.w_lo, w_off[it] + wb, dst + TL_WT);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // Column-major walk of the taps (round 5): the A fragment of halo row r serves tap (dy, dx) of output row r - dy, so for one dx
    // the 4 halo rows x 2 halves a wave touches are read ONCE (16 fragment reads) and feed the three dy taps - 66 ds_read_b128 per
    // chunk and wave instead of 90.  Measured: 315 vs 316 - 321 us per launch at 32 frames (profiles/r05_o_*) - a null: the LDS reads
    // were not the limit.  The launch moves 1.06 GB of pair x 1.33 (halo of the 8 x 32 tile) = 1.41 GB in 0.315 ms = 4.5 TB/s, the
    // rate of this chip's bandwidth kernels: it is HBM-bound on its halo re-reads (a 16 x 32 tile would need 96 KB of LDS per block
    // and lose the second resident block that covers a block's load phase).
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      half8 ah[8], al[8];                        // [halo row 2*wave + r][x half]
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = (2 * wave + r) * TL_HX + h * 16 + dx + fr;
          const int ao = row * TL_BK + ((kq ^ tl_swz(row)) * 8);
          ah[r * 2 + h] = *reinterpret_cast<const half8*>(&s_act[0][ao]);
          al[r * 2 + h] = *reinterpret_cast<const half8*>(&s_act[1][ao]);
        }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int wrow = (dy * 3 + dx) * 16 + fr;
        const int wo = wrow * TL_BK + ((kq ^ tl_swz(wrow)) * 8);
        const half8 bh = *reinterpret_cast<const half8*>(&s_w[0][wo]);
        const half8 bl = *reinterpret_cast<const half8*>(&s_w[1][wo]);
        // M-tile m = output row 2*wave + (m >> 1), x half m & 1 reads halo row (m >> 1) + dy.  Pass-major order (see
        // convhalo.hip): dependent MFMAs (the two acc_x terms of a tile) 4 instructions apart
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc_m[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[((m >> 1) + dy) * 2 + (m & 1)], bh, acc_m[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc_x[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[((m >> 1) + dy) * 2 + (m & 1)], bl, acc_x[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc_x[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[((m >> 1) + dy) * 2 + (m & 1)], bh, acc_x[m], 0, 0, 0);
      }
    }
  }
  // D: lane (class fr, kq) holds pixels 4*kq .. 4*kq + 3 of each M-tile
  if (fr < p.K) {
    const float sc_in = ff3d_pow2(ff3d_ld_exp(p.sc.a_exp) + ff3d_ld_exp(p.sc.w_exp));
    const float bj = p.bias ? p.bias[fr] : 0.f;
    float* o = p.out + ((long long)b * p.K + fr) * p.H * p.W;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int y = ty0 + 2 * wave + (m >> 1);
      if (y >= p.H) continue;
      const int x = tx0 + (m & 1) * 16 + kq * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaf(acc_m[m][r] + acc_x[m][r] * (1.f / 2048.f), sc_in, bj);
      if (x + 3 < p.W && (p.W & 3) == 0) {
        *reinterpret_cast<float4*>(o + (long long)y * p.W + x) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (x + r < p.W) o[(long long)y * p.W + x + r] = v[r];
      }
    }
  }
}

__global__ __launch_bounds__(256, 2) void conv3x3_small_f16x3_kernel(TailParams p) { tl_body(p, blockIdx.x, gridDim.x); }

// Several tail convolutions of one shape in one launch (the heatmap heads of the multi-stage head; see convhalo.hip): at 4 frames
// one is 552 blocks on 512 resident slots - two rounds, the second 8 % full; three together are 3.2 rounds.
constexpr int TL_MAX_GROUP = 4;
struct TailGroup {
  TailParams p[TL_MAX_GROUP];
  unsigned per;
};
__global__ __launch_bounds__(256, 2) void conv3x3_small_group_kernel(TailGroup gp) {
  const unsigned g = blockIdx.x / gp.per;
  tl_body(gp.p[g], blockIdx.x - g * gp.per, gp.per);
}

}  // namespace

extern "C" int ff3d_conv3x3_small_f16x3_group(int n, const void* const* x_hi, const void* const* x_lo, const void* const* w_hi,
                                              const void* const* w_lo, const float* const* bias, float* const* out, int B,
                                              int C, int H, int W, int K, const ff3d_scale_t* const* scale_host,
                                              ff3d_stream_t stream) {
  FF3D_REQUIRE(n >= 1 && n <= TL_MAX_GROUP && x_hi && x_lo && w_hi && w_lo && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && C > 0 && C % TL_BK == 0 && H > 0 && W > 0 && K > 0 && K <= 16, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(((long long)B * H * W + 1) * C * 2 < (1ll << 32), FF3D_ERR_BAD_SHAPE);
  const long long per = (long long)B * ((H + TL_Y - 1) / TL_Y) * ((W + TL_X - 1) / TL_X);
  FF3D_REQUIRE(per * n < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  TailGroup gp{};
  gp.per = (unsigned)per;
  for (int g = 0; g < n; ++g) {
    FF3D_REQUIRE(x_hi[g] && x_lo[g] && w_hi[g] && w_lo[g] && out[g], FF3D_ERR_NULL);
    gp.p[g] = TailParams{static_cast<const _Float16*>(x_hi[g]), static_cast<const _Float16*>(x_lo[g]),
                         static_cast<const _Float16*>(w_hi[g]), static_cast<const _Float16*>(w_lo[g]), bias ? bias[g] : nullptr,
                         out[g], B, C, H, W, K, (unsigned)((long long)B * H * W * C * 2),
                         ff3d_scale_from(scale_host ? scale_host[g] : nullptr)};
  }
  ff3d_clear_error();
  hipLaunchKernelGGL(conv3x3_small_group_kernel, dim3((unsigned)(per * n)), dim3(256), 0, static_cast<hipStream_t>(stream), gp);
  return ff3d_launch_status();
}

extern "C" int ff3d_conv3x3_small_f16x3(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo,
                                        const float* bias, float* out, int B, int C, int H, int W, int K,
                                        const ff3d_scale_t* scale_host, ff3d_stream_t stream) {
  FF3D_REQUIRE(x_hi && x_lo && w_hi && w_lo && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && C > 0 && C % TL_BK == 0 && H > 0 && W > 0 && K > 0 && K <= 16, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(((long long)B * H * W + 1) * C * 2 < (1ll << 32), FF3D_ERR_BAD_SHAPE);   // 32-bit DMA byte offsets
  const long long blocks = (long long)B * ((H + TL_Y - 1) / TL_Y) * ((W + TL_X - 1) / TL_X);
  FF3D_REQUIRE(blocks < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  TailParams p{static_cast<const _Float16*>(x_hi), static_cast<const _Float16*>(x_lo),
               static_cast<const _Float16*>(w_hi), static_cast<const _Float16*>(w_lo), bias, out, B, C, H, W, K,
               (unsigned)((long long)B * H * W * C * 2), ff3d_scale_from(scale_host)};
  ff3d_clear_error();
  hipLaunchKernelGGL(conv3x3_small_f16x3_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return ff3d_launch_status();
}

// Round 5: the same convolution with the class-padded weight planes in CHUNK TILES, wt[c0 / 32][tap][class 0 .. 15][32] = w[class][tap][c0 ..
// c0 + 31]: what a block stages per 32-channel chunk and plane - 9 taps x 16 classes x 64 bytes - is one contiguous 9 216-byte run in LDS
// row order instead of 144 pieces of 64 bytes C * 2 bytes apart, each half of a 128-byte line whose other half is the NEXT chunk's
// (4.7 GB of such pieces L2 -> LDS per 32-frame launch).  Bit-identical results.
extern "C" int ff3d_conv3x3_small_f16x3_tiled(const void* x_hi, const void* x_lo, const void* wt_hi, const void* wt_lo,
                                        const float* bias, float* out, int B, int C, int H, int W, int K,
                                        const ff3d_scale_t* scale_host, ff3d_stream_t stream) {
  FF3D_REQUIRE(x_hi && x_lo && wt_hi && wt_lo && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && C > 0 && C % TL_BK == 0 && H > 0 && W > 0 && K > 0 && K <= 16, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(((long long)B * H * W + 1) * C * 2 < (1ll << 32), FF3D_ERR_BAD_SHAPE);   // 32-bit DMA byte offsets
  const long long blocks = (long long)B * ((H + TL_Y - 1) / TL_Y) * ((W + TL_X - 1) / TL_X);
  FF3D_REQUIRE(blocks < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  TailParams p{static_cast<const _Float16*>(x_hi), static_cast<const _Float16*>(x_lo),
               static_cast<const _Float16*>(wt_hi), static_cast<const _Float16*>(wt_lo), bias, out, B, C, H, W, K,
               (unsigned)((long long)B * H * W * C * 2), ff3d_scale_from(scale_host)};
  p.w_tiled = 1;
  ff3d_clear_error();
  hipLaunchKernelGGL(conv3x3_small_f16x3_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return ff3d_launch_status();
}
