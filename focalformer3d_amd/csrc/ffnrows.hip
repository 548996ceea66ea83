// Round 5: the feed-forward step of a decoder layer (mmcv FFN + the identity add + the post-norm that follows it in
// BaseTransformerLayer's operation order; reached from FD:927-933 through DeformableDetrTransformerDecoder) as ONE launch,
//     out = LayerNorm(residual + relu(x W1^T + b1) W2^T + b2) * gamma + beta        (and out_pos = out + pos),
// x, residual, out (M, 256) fp32, W1 (Hd, 256), W2 (256, Hd), Hd = 1024 in every shipped config - in the split-fp16 arithmetic of
// linear.hip / linrows.hip (every operand a (hi, lo) fp16 pair, three v_mfma_f32_16x16x32_f16 passes, fp32 accumulation) with the
// low parts UNSCALED here: after range normalisation the row maximum sits in [2^13, 2^14), lo = x' - hi is a normal fp16 number
// for every element within 2^-16 of it, and the three passes add into ONE accumulator (lo * hi, hi * lo, hi * hi in that order) -
// half the accumulator registers of the (main, scaled cross terms) pair the other kernels keep.  The Hd-wide hidden activation never
// leaves the CU.
// Why.  Rounds 3-5 ran this step as two launches (linear 19 200 x 256 x 1024: 64 us, then [projection + add + LayerNorm] 19 200 x
// 1024 x 256 on linrows.hip: 71 us): 157 MB of hidden activation written and read back per layer, and two kernels that are each a
// chain of latencies - a block streams its weights through a barrier-stepped LDS ring and drains the queue at every 128-wide K
// half-chunk (profiles/r05_p_*: the ring itself sustains 15 TB/s over the chip; it is the waits that cost).
// Shape.  One 512-thread block (8 waves, one block per CU) owns BM = 16 * MT rows (19 200 rows: MT = 5, 240 blocks, one round):
//   * x: read once; a thread holds its share of the (BM x 256) panel in registers (2 * MT float4), row maximum by xor-shuffles
//     inside the half-wave, ONE power-of-two normalisation per row over all 256 columns, and the whole (hi, lo) image of the
//     panel (8 K-steps) is written to LDS once: BM KiB;
//   * the hidden dimension is walked in chunks of 128 units.  Phase 1 (per chunk): h = relu(x W1[chunk]^T + b1), 8 K-steps from
//     the x image; the waves are 4 column groups (32 units) x 2 row groups (ceil(MT / 2) and floor(MT / 2) row tiles - wave w and
//     w + 4 share a SIMD).  The row maximum of the chunk goes through one LDS exchange, the chunk is normalised per row and
//     written as the (hi, lo) image of phase 2's activation operand (4 K-steps, BM / 2 KiB): a column group's 32 units ARE one
//     K-step.  Phase 2 (per chunk): y += h W2[:, chunk]^T, 4 K-steps; waves = 8 x 32 output columns x all rows (linrows.hip's
//     layout, so the LayerNorm epilogue is the same code), folded into the running fp32 sum with 2^(e_h(row) + e_w2);
//   * weights never touch LDS: a wave reads only ITS rows of a weight tile (the waves split the output columns), so the LDS ring of
//     linrows.hip bought no reuse.  The planes arrive K-STEP-TILED from the host ([K / 32][N][32] halves: ops.tile_weight_f16), a
//     fragment load of a wave is then one contiguous 1 KiB global_load_dwordx4 per (tile, plane), issued TWO weight steps ahead
//     into a 3-deep register ring - also across the phase and chunk boundaries - and waited for by the compiler's own vmcnt
//     bookkeeping.  No barrier inside a K loop: the LDS images are static while they are read; 2 barriers per chunk;
//   * first version (two accumulators, one step ahead): 256 VGPRs + 31 spilled at MT = 5 - the reloads sit behind the weight loads
//     in the in-order VM counter - 105 us per launch against 124 for the two launches; block time 35 us + 9 us x MT at MT = 2 .. 4
//     (profiles/r05_s_*): the fixed part is weight latency, hence the deeper ring and the single accumulator that pays for it;
//   * LDS: BM KiB (x) + BM / 2 KiB (h) + 4 KiB = 124 KiB at MT = 5; 61 440 MFMA cycles per SIMD and block = 26 us at 2.4 GHz.
#include <cstdlib>

#include "ff3d_common.h"

namespace {

using half4 = __attribute__((ext_vector_type(4))) _Float16;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int FF_BK = 32, FF_C = 256, FF_T = 512, FF_HC = 128;    // K-step, model width, threads, hidden units per chunk

__device__ __forceinline__ int ff_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

struct FfnParams {
  const float* x;
  long long lda;
  const _Float16 *w1_hi, *w1_lo;   // [8][Hd][32] K-step tiles of W1 (Hd, 256)
  const _Float16 *w2_hi, *w2_lo;   // [Hd / 32][256][32] K-step tiles of W2 (256, Hd)
  const int *w1_exp, *w2_exp;
  const float *b1, *b2;
  const float *res, *gamma, *beta, *pos;
  float *out, *out2;
  float eps;
  int M, Hd;
};

__device__ __forceinline__ int ff_row_exp(float mx) {          // max * 2^-e in [2^13, 2^14); zero rows keep e = 0
  const int eb = (int)((__float_as_uint(mx) >> 23) & 0xffu);
  return (mx > 0.f) ? eb - 127 - 13 : 0;
}

template <int MT>
__global__ __launch_bounds__(FF_T, 1) void ffn_rows_kernel(FfnParams p) {
  constexpr int BM = 16 * MT, NT = 2;
  constexpr int R0 = (MT + 1) / 2;                                  // row tiles of row group 0; group 1 has MT - R0
  constexpr int A_TILE = BM * FF_BK, A_STEP = 2 * A_TILE;           // halves: one plane tile / one K-step (both planes)
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];    // x image [8][2][A]  h image [4][2][A]  exps / maxima / LN sums
  _Float16* const ximg = lds;
  _Float16* const himg = lds + 8 * A_STEP;
  int* const s_xexp = reinterpret_cast<int*>(himg + 4 * A_STEP);
  int* const s_hexp = s_xexp + BM;
  float* const s_hmax = reinterpret_cast<float*>(s_hexp + BM);      // [4 column groups][BM]
  float* const s_red = s_hmax + 4 * BM;                              // [2][BM][8]
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: the row-group predicates below are s_cbranch, not exec masks
  const unsigned lid = ff3d_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (int)lid * BM;
  const int Hd = p.Hd, nchunks = Hd / FF_HC;

  // ---- weights: a lane's fragment of K-step tile `tile` row block `r0` is 16 bytes at ((tile * rows + r0 + fr) * 32 + kq * 8) halves.
  //      One chunk = 12 weight steps: 0-7 = W1 (8 K-steps over the 256 inputs, rows = this column group's 32 units), 8-11 = W2 (the
  //      chunk's 4 K-steps, rows = this wave's 32 output columns).  A 3-deep register ring holds the step being used and the two
  //      after it (12 % 3 == 0: static ring slots in the unrolled chunk body); the loads are plain global loads, the compiler's
  //      vmcnt bookkeeping waits for the oldest only.
  const int cg = wave & 3, rg = wave >> 2;
  const int nrt = rg ? MT - R0 : R0, rt0 = rg ? R0 : 0;             // this wave's row tiles in phase 1
  const unsigned w_lane = (unsigned)(fr * FF_BK + kq * 8);           // (the only per-lane part: everything else is a scalar base)
  half8 wh[3][NT], wl[3][NT];
  auto fetch = [&](int c, int g, int slot) {                         // weight step g (0 .. 11) of chunk c -> ring slot
    if (c >= nchunks) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (g < 8) {
        const long long o = ((long long)g * Hd + c * FF_HC + cg * 32 + t * 16) * FF_BK;
        wh[slot][t] = *reinterpret_cast<const half8*>(p.w1_hi + o + w_lane);
        wl[slot][t] = *reinterpret_cast<const half8*>(p.w1_lo + o + w_lane);
      } else {
        const long long o = ((long long)(c * 4 + g - 8) * FF_C + wave * 32 + t * 16) * FF_BK;
        wh[slot][t] = *reinterpret_cast<const half8*>(p.w2_hi + o + w_lane);
        wl[slot][t] = *reinterpret_cast<const half8*>(p.w2_lo + o + w_lane);
      }
    }
  };
  fetch(0, 0, 0);
  fetch(0, 1, 1);

  // ---- x panel -> its (hi, lo) image, one exponent per row.  Thread -> float4 a_c4 of half hc of rows j * 16 + a_r
  {
    const int a_r = tid >> 5, a_c4 = tid & 31;
    const int a_ks = a_c4 >> 3, a_q = (a_c4 & 7) >> 1, a_sub = (a_c4 & 1) * 4;
    f32x4 ra[2][MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int row = j * 16 + a_r;
      const float* src = p.x + (long long)min(m0 + row, p.M - 1) * p.lda + a_c4 * 4;
#pragma unroll
      for (int hc = 0; hc < 2; ++hc) ra[hc][j] = *reinterpret_cast<const f32x4*>(src + hc * 128);
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int row = j * 16 + a_r;
      const bool real = m0 + row < p.M;
      float v[2][4];
      float mx = 0.f;
#pragma unroll
      for (int hc = 0; hc < 2; ++hc)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[hc][i] = real ? ra[hc][j][i] : 0.f;
          mx = fmaxf(mx, fabsf(v[hc][i]));
        }
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));        // the row's 32 lanes = this half-wave
      const int e = ff_row_exp(mx);
      const float inv = ff3d_pow2(-e);
      if (a_c4 == 0) s_xexp[row] = e;
#pragma unroll
      for (int hc = 0; hc < 2; ++hc) {
        half4 hh, ll;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x_ = v[hc][i] * inv;
          const _Float16 h_ = (_Float16)x_;
          hh[i] = h_;
          ll[i] = (_Float16)(x_ - (float)h_);
        }
        const int o = (hc * 4 + a_ks) * A_STEP + row * FF_BK + ((a_q ^ ff_swz(row)) * 8) + a_sub;
        *reinterpret_cast<half4*>(ximg + o) = hh;
        *reinterpret_cast<half4*>(ximg + o + A_TILE) = ll;
      }
    }
  }
  __syncthreads();

  const int we1 = ff3d_ld_exp(p.w1_exp), we2 = ff3d_ld_exp(p.w2_exp);
  // LDS offsets (halves) of this lane inside a plane tile: the swizzle of row m * 16 + fr depends on fr only, so a row tile is a
  // constant displacement of 16 * 32 halves from the lane's base
  const int a_base = fr * FF_BK + ((kq ^ ff_swz(fr)) * 8);
  int h_base[NT];                                                   // where this lane's 4 units of tile t go in the h image
#pragma unroll
  for (int t = 0; t < NT; ++t) h_base[t] = fr * FF_BK + (((2 * t + (kq >> 1)) ^ ff_swz(fr)) * 8) + (kq & 1) * 4;
  f32x4 sum[NT][MT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m) sum[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c = 0; c < nchunks; ++c) {
    // ---------------- phase 1: h (rows of this row group) x (32 units of this column group) over K = 256
    f32x4 hm[NT][R0];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int m = 0; m < R0; ++m) hm[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      fetch(ks + 2 < 12 ? c : c + 1, (ks + 2) % 12, (ks + 2) % 3);
      const _Float16* A = ximg + ks * A_STEP;
      half8 ah[R0], al[R0];
#pragma unroll
      for (int m = 0; m < R0; ++m) {
        if (m < nrt) {
          const int o = a_base + (rt0 + m) * (16 * FF_BK);
          ah[m] = *reinterpret_cast<const half8*>(A + o);
          al[m] = *reinterpret_cast<const half8*>(A + A_TILE + o);
        }
      }
      // ONE accumulator for the three passes (the low planes are unscaled), pass-major (convhalo.hip): two MFMAs on the same
      // accumulator are 2 * R0 instructions apart
#pragma unroll
      for (int m = 0; m < R0; ++m)
        if (m < nrt)
#pragma unroll
          for (int t = 0; t < NT; ++t) hm[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ks % 3][t], ah[m], hm[t][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < R0; ++m)
        if (m < nrt)
#pragma unroll
          for (int t = 0; t < NT; ++t) hm[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ks % 3][t], al[m], hm[t][m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < R0; ++m)
        if (m < nrt)
#pragma unroll
          for (int t = 0; t < NT; ++t) hm[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ks % 3][t], ah[m], hm[t][m], 0, 0, 0);
    }
    // lane (fr, kq): units u0 + t * 16 + 4 kq .. + 3 of row (rt0 + m) * 16 + fr; bias, ReLU, the wave's share of the row maximum
    const int u0 = c * FF_HC + cg * 32;
#pragma unroll
    for (int m = 0; m < R0; ++m) {
      if (m < nrt) {
        const int row = (rt0 + m) * 16 + fr;
        const float sc_f = ff3d_pow2(s_xexp[row] + we1);
        float mx = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 b = *reinterpret_cast<const float4*>(p.b1 + u0 + t * 16 + 4 * kq);
          const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float h = fmaxf(fmaf(hm[t][m][i], sc_f, bb[i]), 0.f);
            hm[t][m][i] = h;
            mx = fmaxf(mx, h);
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (kq == 0) s_hmax[cg * BM + row] = mx;
      }
    }
    __syncthreads();            // the four column groups' maxima are in; every wave is past phase 2 of the previous chunk (h image free)
#pragma unroll
    for (int m = 0; m < R0; ++m) {
      if (m < nrt) {
        const int row = (rt0 + m) * 16 + fr;
        const float mx = fmaxf(fmaxf(s_hmax[row], s_hmax[BM + row]), fmaxf(s_hmax[2 * BM + row], s_hmax[3 * BM + row]));
        const int e = ff_row_exp(mx);
        const float inv = ff3d_pow2(-e);
        if (cg == 0 && kq == 0) s_hexp[row] = e;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          half4 hh, ll;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float x_ = hm[t][m][i] * inv;
            const _Float16 h_ = (_Float16)x_;
            hh[i] = h_;
            ll[i] = (_Float16)(x_ - (float)h_);
          }
          // unit t * 16 + 4 kq + i of the group = column of K-step cg: 16-byte chunk 2 t + (kq >> 1), halves (kq & 1) * 4 ..
          const int o = cg * A_STEP + (rt0 + m) * (16 * FF_BK) + h_base[t];
          *reinterpret_cast<half4*>(himg + o) = hh;
          *reinterpret_cast<half4*>(himg + o + A_TILE) = ll;
        }
      }
    }
    __syncthreads();            // the h image and its exponents are visible

    // ---------------- phase 2: y (all rows) x (32 columns of this wave) += h W2[:, chunk]^T over the chunk's 4 K-steps
    f32x4 am[NT][MT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int m = 0; m < MT; ++m) am[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fetch(ks + 10 < 12 ? c : c + 1, (ks + 10) % 12, (ks + 10) % 3);
      const _Float16* A = himg + ks * A_STEP;
      constexpr int S = (8 + 0) % 3;                                   // ring slot of weight step 8 + ks = (S + ks) % 3
      // row tiles in two groups (R0, then the rest): the activation fragments of a group are 8 * R0 registers instead of 8 * MT,
      // pass-major inside a group
#pragma unroll
      for (int g0 = 0; g0 < MT; g0 += R0) {
        half8 ah[R0], al[R0];
#pragma unroll
        for (int m = 0; m < R0; ++m)
          if (g0 + m < MT) {
            const int o = a_base + (g0 + m) * (16 * FF_BK);
            ah[m] = *reinterpret_cast<const half8*>(A + o);
            al[m] = *reinterpret_cast<const half8*>(A + A_TILE + o);
          }
#pragma unroll
        for (int m = 0; m < R0; ++m)
          if (g0 + m < MT)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              am[t][g0 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[(S + ks) % 3][t], ah[m], am[t][g0 + m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < R0; ++m)
          if (g0 + m < MT)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              am[t][g0 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(S + ks) % 3][t], al[m], am[t][g0 + m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < R0; ++m)
          if (g0 + m < MT)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              am[t][g0 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(S + ks) % 3][t], ah[m], am[t][g0 + m], 0, 0, 0);
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float sc_f = ff3d_pow2(s_hexp[m * 16 + fr] + we2);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) sum[t][m][i] = fmaf(am[t][m][i], sc_f, sum[t][m][i]);
    }
    // (the next chunk's first barrier orders these reads of s_hexp / the h image before they are written again)
  }

  // ---- epilogue (linrows.hip's): lane (fr, kq) holds columns wave * 32 + t * 16 + 4 kq .. + 3 of row m0 + m * 16 + fr in sum[t][m]
  const float inv_n = 1.f / (float)FF_C;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int gm = min(m0 + m * 16 + fr, p.M - 1);
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = wave * 32 + t * 16 + 4 * kq;
      const float4 b = p.b2 ? *reinterpret_cast<const float4*>(p.b2 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 r = *reinterpret_cast<const float4*>(p.res + (long long)gm * FF_C + n);
      sum[t][m][0] = (sum[t][m][0] + b.x) + r.x, sum[t][m][1] = (sum[t][m][1] + b.y) + r.y;
      sum[t][m][2] = (sum[t][m][2] + b.z) + r.z, sum[t][m][3] = (sum[t][m][3] + b.w) + r.w;
      s1 += (sum[t][m][0] + sum[t][m][1]) + (sum[t][m][2] + sum[t][m][3]);
    }
    s1 += __shfl_xor(s1, 16);
    s1 += __shfl_xor(s1, 32);
    if (kq == 0) s_red[(m * 16 + fr) * 8 + wave] = s1;
  }
  __syncthreads();
  float mean[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float4 r0 = *reinterpret_cast<const float4*>(s_red + (m * 16 + fr) * 8);
    const float4 r1 = *reinterpret_cast<const float4*>(s_red + (m * 16 + fr) * 8 + 4);
    mean[m] = (((r0.x + r0.y) + (r0.z + r0.w)) + ((r1.x + r1.y) + (r1.z + r1.w))) * inv_n;
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float d = sum[t][m][i] - mean[m];
        s2 = fmaf(d, d, s2);
      }
    s2 += __shfl_xor(s2, 16);
    s2 += __shfl_xor(s2, 32);
    if (kq == 0) s_red[BM * 8 + (m * 16 + fr) * 8 + wave] = s2;
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float4 r0 = *reinterpret_cast<const float4*>(s_red + BM * 8 + (m * 16 + fr) * 8);
    const float4 r1 = *reinterpret_cast<const float4*>(s_red + BM * 8 + (m * 16 + fr) * 8 + 4);
    const float rstd = rsqrtf((((r0.x + r0.y) + (r0.z + r0.w)) + ((r1.x + r1.y) + (r1.z + r1.w))) * inv_n + p.eps);
    const int gm = m0 + m * 16 + fr;
    if (gm >= p.M) continue;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = wave * 32 + t * 16 + 4 * kq;
      const float4 g = *reinterpret_cast<const float4*>(p.gamma + n), be = *reinterpret_cast<const float4*>(p.beta + n);
      float4 y;
      y.x = (sum[t][m][0] - mean[m]) * rstd * g.x + be.x;
      y.y = (sum[t][m][1] - mean[m]) * rstd * g.y + be.y;
      y.z = (sum[t][m][2] - mean[m]) * rstd * g.z + be.z;
      y.w = (sum[t][m][3] - mean[m]) * rstd * g.w + be.w;
      *reinterpret_cast<float4*>(p.out + (long long)gm * FF_C + n) = y;
      if (p.out2) {
        const float4 q = *reinterpret_cast<const float4*>(p.pos + (long long)gm * FF_C + n);
        *reinterpret_cast<float4*>(p.out2 + (long long)gm * FF_C + n) = make_float4(y.x + q.x, y.y + q.y, y.z + q.z, y.w + q.w);
      }
    }
  }
}

template <int MT>
int launch_ffn(const FfnParams& p, hipStream_t s) {
  constexpr int BM = 16 * MT;
  constexpr size_t lds_bytes = (size_t)12 * 2 * BM * FF_BK * sizeof(_Float16) + 2 * BM * sizeof(int) + 4 * BM * sizeof(float) +
                               2 * BM * 8 * sizeof(float);
  static_assert(lds_bytes <= 160 * 1024, "LDS budget");
  static bool configured[64] = {};                // > 64 KiB of dynamic LDS has to be enabled once per kernel AND device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!configured[dev & 63]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_rows_kernel<MT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess)
      return FF3D_ERR_LAUNCH;
    configured[dev & 63] = true;
  }
  const int blocks = (p.M + BM - 1) / BM;
  ff3d_clear_error();
  hipLaunchKernelGGL((ffn_rows_kernel<MT>), dim3((unsigned)blocks), dim3(FF_T), lds_bytes, s, p);
  return ff3d_launch_status();
}

// Rows per block: the height (2 .. 5 row tiles) whose grid needs the least (rounds of the chip) x (time of a block ~ 1 + MT: the x
// image and the weight stream of a block do not depend on its height); ties -> the taller block.  FF3D_FFN_MT forces one (A/B runs).
int ffn_mt(int M) {
  static const int forced = [] {
    const char* e = getenv("FF3D_FFN_MT");
    return e ? atoi(e) : 0;
  }();
  if (forced >= 2 && forced <= 5) return forced;
  static int cus[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!cus[dev & 63]) {
    hipDeviceProp_t prop;
    cus[dev & 63] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const long long slots = cus[dev & 63];
  int best = 2;
  long long best_cost = -1;
  for (int mt = 2; mt <= 5; ++mt) {
    const long long blocks = (M + 16 * mt - 1) / (16 * mt);
    const long long cost = ((blocks + slots - 1) / slots) * (1 + mt);
    if (best_cost < 0 || cost <= best_cost) best = mt, best_cost = cost;
  }
  return best;
}

}  // namespace

extern "C" int ff3d_ffn_rows(const float* x, int64_t lda, const void* w1t_hi, const void* w1t_lo, const int32_t* w1_exp,
                             const float* b1, int hidden, const void* w2t_hi, const void* w2t_lo, const int32_t* w2_exp,
                             const float* b2, const float* residual, const float* gamma, const float* beta, float eps,
                             const float* pos, float* out, float* out_pos, int M, ff3d_stream_t stream) {
  FF3D_REQUIRE(x && w1t_hi && w1t_lo && w2t_hi && w2t_lo && b1 && residual && gamma && beta && out && (!out_pos || pos), FF3D_ERR_NULL);
  FF3D_REQUIRE(M > 0 && hidden > 0 && hidden % FF_HC == 0 && hidden <= 65536 && lda >= FF_C, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(x) && lda % 4 == 0 && ff3d_aligned16(w1t_hi) && ff3d_aligned16(w1t_lo) && ff3d_aligned16(w2t_hi) &&
                   ff3d_aligned16(w2t_lo) && ff3d_aligned16(b1) && (!b2 || ff3d_aligned16(b2)) && ff3d_aligned16(residual) &&
                   ff3d_aligned16(gamma) && ff3d_aligned16(beta) && ff3d_aligned16(out) && (!pos || ff3d_aligned16(pos)) &&
                   (!out_pos || ff3d_aligned16(out_pos)),
               FF3D_ERR_ALIGNMENT);
  FfnParams p{};
  p.x = x, p.lda = lda;
  p.w1_hi = static_cast<const _Float16*>(w1t_hi), p.w1_lo = static_cast<const _Float16*>(w1t_lo), p.w1_exp = w1_exp, p.b1 = b1;
  p.w2_hi = static_cast<const _Float16*>(w2t_hi), p.w2_lo = static_cast<const _Float16*>(w2t_lo), p.w2_exp = w2_exp, p.b2 = b2;
  p.res = residual, p.gamma = gamma, p.beta = beta, p.pos = pos, p.out = out, p.out2 = out_pos, p.eps = eps, p.M = M, p.Hd = hidden;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (ffn_mt(M)) {
    case 2: return launch_ffn<2>(p, s);
    case 3: return launch_ffn<3>(p, s);
    case 4: return launch_ffn<4>(p, s);
    default: return launch_ffn<5>(p, s);
  }
}
