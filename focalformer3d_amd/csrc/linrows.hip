// Round 5: the query-side projections of the decoder (linear.hip's family; reached from FD:870-871, 914-922, 927-933) as a
// ROW-OWNING kernel for the sizes where linear.hip's tiles quantise badly, in two arithmetics:
//     out (M, N) fp32 = epilogue(A (M, K) fp32 @ W (N, K)^T + bias)
//   PLANES = 2  split-fp16 (fp32-class: (hi, lo') fp16 pairs, three v_mfma_f32_16x16x32_f16 passes, per-row power-of-two
//               normalisation inside the kernel - linear.hip's arithmetic);
//   PLANES = 1  bf16 operands on v_mfma_f32_16x16x32_bf16 (BASELINE.json configs[4]: "bf16 QKV/FFN on MFMA"): activation
//               rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while it is staged, weights a bf16 plane, fp32 accumulation, bias added
//               in fp32, ONE rounding of the result to bf16, ReLU on the rounded value, result stored as fp32 - exactly
//               oracle/ff3d_oracle.py lin(lowp=True).  Rounds 1-4 ran this mode as F.linear on hipBLASLt + two cast launches per
//               projection (77 cast launches and a stream-K GEMM that forbids overlapping replays in the configs[4] step).
// Why another tiling.  At 32 frames (M = 19 200 rows) linear.hip's 64 x 128 tiles make 600 blocks for 512 resident slots - 1.17
// rounds, the second 17 % full - and the fused [projection + residual + LayerNorm] form (a block must own whole 256-column rows)
// streams all of W per 32 rows: 68 us against 34 + 20 for two launches (profiles/r03_n_*), so rounds 3-4 ran 18 LayerNorm launches
// beside 44 projections at 0.16 MFMA-busy.  Here a block is 8 waves (512 threads, one block per CU in the split mode) and owns
// BM = 16 * MT rows x 256 columns, MT chosen on the host so that the grid is ONE full round of the chip where possible
// (19 200 rows: MT = 5, 240 blocks); wave = all BM rows x 32 columns (NT = 2: the A fragments a wave reads serve two column tiles).
//   * activation: K is walked in HALF-chunks of 128 (4 K-steps).  A thread holds MT float4 of the block's (BM x 128) fp32 panel
//     (row j * 16 + tid / 32, floats 4 * (tid % 32) ..: a half-wave reads one row's 512 contiguous bytes), issued a half-chunk
//     ahead by inline-asm loads; after they land: row maximum by xor-shuffles inside the half-wave, per-row power-of-two
//     normalisation (PLANES = 2), conversion, and the WHOLE half-chunk image ((hi, lo') or bf16 tiles of its 4 K-steps) is written
//     to LDS at once - the K-steps then run from LDS with no VALU work between the MFMAs;
//   * weights: 256 x 32 tiles per plane by 16-byte LDS DMA into a 3-stage ring, two K-steps ahead; the only counted wait is
//     vmcnt(PW) (the newest step stays in flight across the barrier); a half-chunk boundary drains the queue (it needs the
//     activation loads, which are older than the weights in flight);
//   * LDS: 4 x PLANES x BM x 64 B (image) + 3 x PLANES x 16 KiB (ring): 136 KiB at MT = 5, PLANES = 2; 68 KiB at PLANES = 1 (two
//     blocks per CU);
//   * epilogues: bias (+ bf16 rounding) + ReLU -> fp32 rows at any stride; or the decoder layer's post-norm step
//     LayerNorm(residual + .) (+ query_pos as a second output) over the 256 columns the block owns (N = 256): row sums through
//     shuffles + one LDS exchange between the 8 waves, two-pass variance as ops.add_layer_norm computes it;
//   * dual activation (column tiles from n_split on read a2): q | k from x + pos and v from x in one launch.
#include <cstdlib>

#include "ff3d_common.h"

namespace {

using half4 = __attribute__((ext_vector_type(4))) _Float16;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int LR_BK = 32, LR_HC = 4, LR_BN = 256, LR_NSTG = 3, LR_T = 512;

__device__ __forceinline__ int lr_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

struct RowsParams {
  const float *a, *a2;          // a2: activation of the column tiles n0 >= n_split (same lda), or null
  int n_split;
  const float *res, *gamma, *beta, *pos;   // LN: residual (M, 256), LayerNorm affine, optional second output out2 = y + pos
  float* out2;
  float eps;
  const _Float16 *w_hi, *w_lo;  // PLANES = 1: w_hi is the bf16 plane, w_lo unused
  const int* w_exp;
  const float* bias;
  float* out;
  long long lda, ldc;
  int M, N, K, act;
};

__device__ __forceinline__ void lr_glds16(const _Float16* base, unsigned byte_off, _Float16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(base) + byte_off,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float lr_round_bf16(float v) { return (float)(__bf16)v; }

template <int MT, int PLANES, bool LN>
__global__ __launch_bounds__(LR_T, PLANES == 1 ? 2 : 1) void linear_rows_kernel(RowsParams p) {
  constexpr int BM = 16 * MT, NT = 2;
  constexpr int A_TILE = BM * LR_BK, A_STEP = PLANES * A_TILE;       // halves: one plane tile / one K-step of the image
  constexpr int B_TILE = LR_BN * LR_BK, W_STAGE = PLANES * B_TILE;   // halves
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];      // [4 steps][planes][A]  [3][planes][W]  exp[BM]  red[2][BM][8]
  _Float16* const ldsW = lds + LR_HC * A_STEP;
  int* const s_exp = reinterpret_cast<int*>(ldsW + LR_NSTG * W_STAGE);
  float* const s_red = reinterpret_cast<float*>(s_exp + BM);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int n_tiles = (p.N + LR_BN - 1) / LR_BN, m_tiles = (p.M + BM - 1) / BM;
  const unsigned lid = ff3d_xcd_remap(blockIdx.x, (unsigned)(n_tiles * m_tiles));
  const int n0 = (int)(lid % n_tiles) * LR_BN, m0 = (int)(lid / n_tiles) * BM;
  const int nk = p.K / LR_BK, nhc = (nk + LR_HC - 1) / LR_HC;

  // ---- activation staging: thread -> float4 a_c4 (of the half-chunk's 32) of rows j * 16 + a_r, j < MT
  const int a_r = tid >> 5, a_c4 = tid & 31;
  const int a_ks = a_c4 >> 3, a_q = (a_c4 & 7) >> 1, a_sub = (a_c4 & 1) * 4;
  const float* const a_base = (p.a2 && n0 >= p.n_split) ? p.a2 : p.a;
  const float* a_ptr[MT];
  bool a_real[MT];
  int a_lds[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int row = j * 16 + a_r;
    a_real[j] = m0 + row < p.M;
    a_ptr[j] = a_base + (long long)min(m0 + row, p.M - 1) * p.lda + (a_c4 & 7) * 4;   // (always a real row: load, select later)
    a_lds[j] = a_ks * A_STEP + row * LR_BK + ((a_q ^ lr_swz(row)) * 8) + a_sub;
  }
  f32x4 ra[MT];
  auto issue_a = [&](int hc) {                                        // MT loads of half-chunk hc (K-step clamped to a real one)
    const int steps = min(LR_HC, nk - hc * LR_HC);
    const int col = (hc * LR_HC + min(a_ks, steps - 1)) * LR_BK;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const float* src = a_ptr[j] + col;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[j]) : "v"(src) : "memory");
    }
  };
  auto convert_a = [&](int hc) {                                      // landed half-chunk -> its LDS image (all of its K-steps)
    const bool on = a_ks < min(LR_HC, nk - hc * LR_HC);
#pragma unroll
    for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(ra[j]));     // uses stay behind the wait that precedes this call
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const bool use = on && a_real[j];
      const float v0 = use ? ra[j][0] : 0.f, v1 = use ? ra[j][1] : 0.f, v2 = use ? ra[j][2] : 0.f, v3 = use ? ra[j][3] : 0.f;
      if (PLANES == 2) {
        float mx = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3)));
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));     // the row's 32 lanes = this half-wave
        // max * 2^-e in [2^13, 2^14); zero rows keep e = 0; a NaN / inf row keeps its NaN / inf through the scaled split
        const int eb = (int)((__float_as_uint(mx) >> 23) & 0xffu);
        const int e = (mx > 0.f) ? eb - 127 - 13 : 0;
        const float inv = ff3d_pow2(-e);
        if (a_c4 == 0) s_exp[j * 16 + a_r] = e;
        half4 hh, ll;
#define LR_SPLIT(i, val)                               \
  {                                                    \
    const float x_ = (val) * inv;                      \
    const _Float16 h_ = (_Float16)x_;                  \
    hh[i] = h_;                                        \
    ll[i] = (_Float16)((x_ - (float)h_) * 2048.f);     \
  }
        LR_SPLIT(0, v0) LR_SPLIT(1, v1) LR_SPLIT(2, v2) LR_SPLIT(3, v3)
#undef LR_SPLIT
        *reinterpret_cast<half4*>(lds + a_lds[j]) = hh;
        *reinterpret_cast<half4*>(lds + a_lds[j] + A_TILE) = ll;
      } else {
        bf16x4 bb;
        bb[0] = (__bf16)v0, bb[1] = (__bf16)v1, bb[2] = (__bf16)v2, bb[3] = (__bf16)v3;
        *reinterpret_cast<bf16x4*>(lds + a_lds[j]) = bb;
      }
    }
  };

  // ---- weight staging: 256 rows x 4 chunks per plane = 1024 16-byte pieces per K-step and plane, 2 per thread (LDS DMA:
  //      lane-linear destination, swizzle applied on the per-lane source address); row N = the plane's zero row
  const _Float16 *w_hi = p.w_hi, *w_lo = p.w_lo;
  unsigned w_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int s = j * LR_T + tid, row = s >> 2, n = n0 + row;
    w_off[j] = (unsigned)min(n, p.N) * (unsigned)p.K * 2u + (unsigned)(((s & 3) ^ lr_swz(row)) * 16);
  }
  auto dma_w = [&](int g) {                                           // K-step g -> ring stage g % 3
    _Float16* base = ldsW + (g % LR_NSTG) * W_STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      _Float16* dst = base + (j * LR_T + wave * 64) * 8;             // wave-uniform: 64 lanes x 16 B behind it
      lr_glds16(w_hi, w_off[j] + (unsigned)g * 64u, dst);
      if (PLANES == 2) lr_glds16(w_lo, w_off[j] + (unsigned)g * 64u, dst + B_TILE);
    }
  };

  f32x4 sum[NT][MT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m) sum[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int we = PLANES == 2 ? ff3d_ld_exp(p.w_exp) : 0;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (the set-up loads are out of the counted window)
  issue_a(0);
  dma_w(0);
  if (nk > 1) dma_w(1);
  for (int hc = 0; hc < nhc; ++hc) {
    const int g0 = hc * LR_HC, steps = min(LR_HC, nk - g0);
    const bool more = hc + 1 < nhc;
    // A(hc) landed - and with it every weight step issued so far: W(g0), W(g0 + 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    convert_a(hc);
    if (more) issue_a(hc + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // the image of hc and W(g0) are visible to every wave
    asm volatile("" ::: "memory");
    f32x4 am[NT][MT], ax[NT][MT];
    if (PLANES == 2) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) am[t][m] = f32x4{0.f, 0.f, 0.f, 0.f}, ax[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < LR_HC; ++ks) {
      if (ks >= steps) break;                                          // block-uniform
      const int g = g0 + ks;
      const bool ahead = g + 2 < nk;
      if (ahead) dma_w(g + 2);                                         // into the stage step g - 1 read (barrier behind it)
      const _Float16* A = lds + ks * A_STEP;
      const _Float16* W = ldsW + (g % LR_NSTG) * W_STAGE;
      if (PLANES == 2) {
        half8 wh[NT], wl[NT], ah[MT], al[MT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int row = wave * 32 + t * 16 + fr;
          const int o = row * LR_BK + ((kq ^ lr_swz(row)) * 8);
          wh[t] = *reinterpret_cast<const half8*>(W + o);
          wl[t] = *reinterpret_cast<const half8*>(W + B_TILE + o);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int row = m * 16 + fr;
          const int o = row * LR_BK + ((kq ^ lr_swz(row)) * 8);
          ah[m] = *reinterpret_cast<const half8*>(A + o);
          al[m] = *reinterpret_cast<const half8*>(A + A_TILE + o);
        }
        // pass-major order (convhalo.hip): the two dependent cross-term MFMAs of a tile are 2 * MT instructions apart
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t) am[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], ah[m], am[t][m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t) ax[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], al[m], ax[t][m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t) ax[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], ah[m], ax[t][m], 0, 0, 0);
      } else {
        bf16x8 wb[NT], ab[MT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int row = wave * 32 + t * 16 + fr;
          wb[t] = *reinterpret_cast<const bf16x8*>(W + row * LR_BK + ((kq ^ lr_swz(row)) * 8));
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int row = m * 16 + fr;
          ab[m] = *reinterpret_cast<const bf16x8*>(A + row * LR_BK + ((kq ^ lr_swz(row)) * 8));
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t) sum[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[t], ab[m], sum[t][m], 0, 0, 0);
      }
      if (ks + 1 < steps) {
        // W(g + 1) landed?  ks = 0: it is older than this half-chunk's drain.  Later: only W(g + 2), just issued, may stay in
        // flight (the activation loads of the next half-chunk are older than W(g + 1) and land with it)
        if (ks > 0) {
          if (ahead) {
            if (PLANES == 2)
              asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else
              asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // this step's fragment reads retired
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    if (PLANES == 2) {
      // fold the half-chunk into the running sum: 2^(row exponent of this half-chunk + weight exponent)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float sc_f = ff3d_pow2(s_exp[m * 16 + fr] + we);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) sum[t][m][i] = fmaf(am[t][m][i] + ax[t][m][i] * (1.f / 2048.f), sc_f, sum[t][m][i]);
      }
    }
    if (more) {                             // every wave is done with the image, the exponents and the last step's weight stage
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }

  // ---- epilogue: lane (fr, kq) holds columns n0 + wave * 32 + t * 16 + 4 kq .. + 3 of row m0 + m * 16 + fr in sum[t][m]
  if (LN) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float inv_n = 1.f / (float)LR_BN;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int gm = min(m0 + m * 16 + fr, p.M - 1);
      float s1 = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n = wave * 32 + t * 16 + 4 * kq;
        const float4 b = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 r = *reinterpret_cast<const float4*>(p.res + (long long)gm * LR_BN + n);
        float y[4] = {sum[t][m][0] + b.x, sum[t][m][1] + b.y, sum[t][m][2] + b.z, sum[t][m][3] + b.w};
        if (PLANES == 1)
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = lr_round_bf16(y[i]);
        sum[t][m][0] = y[0] + r.x, sum[t][m][1] = y[1] + r.y, sum[t][m][2] = y[2] + r.z, sum[t][m][3] = y[3] + r.w;
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
        *reinterpret_cast<float4*>(p.out + (long long)gm * LR_BN + n) = y;
        if (p.out2) {
          const float4 q = *reinterpret_cast<const float4*>(p.pos + (long long)gm * LR_BN + n);
          *reinterpret_cast<float4*>(p.out2 + (long long)gm * LR_BN + n) = make_float4(y.x + q.x, y.y + q.y, y.z + q.z, y.w + q.w);
        }
      }
    }
    return;
  }
  const bool vec = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15u) == 0 &&
                   (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15u) == 0);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int gm = m0 + m * 16 + fr;
    if (gm >= p.M) continue;
    float* orow = p.out + (long long)gm * p.ldc;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + wave * 32 + t * 16 + 4 * kq;
      if (n >= p.N) continue;
      float v[4] = {sum[t][m][0], sum[t][m][1], sum[t][m][2], sum[t][m][3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (p.bias && n + i < p.N) v[i] += p.bias[n + i];
        if (PLANES == 1) v[i] = lr_round_bf16(v[i]);
        if (p.act) v[i] = fmaxf(v[i], 0.f);
      }
      if (vec && n + 3 < p.N) {
        *reinterpret_cast<float4*>(orow + n) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (n + i < p.N) orow[n + i] = v[i];
      }
    }
  }
}

template <int MT, int PLANES, bool LN>
int launch_rows(const RowsParams& p, hipStream_t s) {
  constexpr int BM = 16 * MT;
  constexpr size_t lds_bytes = (size_t)(LR_HC * PLANES * BM * LR_BK + LR_NSTG * PLANES * LR_BN * LR_BK) * sizeof(_Float16) +
                               BM * sizeof(int) + 2 * BM * 8 * sizeof(float);
  static_assert(lds_bytes <= 160 * 1024, "LDS budget");
  static bool configured[64] = {};                // > 64 KiB of dynamic LDS has to be enabled once per kernel AND device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!configured[dev & 63]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_rows_kernel<MT, PLANES, LN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
      return FF3D_ERR_LAUNCH;
    configured[dev & 63] = true;
  }
  const int blocks = ((p.M + BM - 1) / BM) * ((p.N + LR_BN - 1) / LR_BN);
  ff3d_clear_error();
  hipLaunchKernelGGL((linear_rows_kernel<MT, PLANES, LN>), dim3((unsigned)blocks), dim3(LR_T), lds_bytes, s, p);
  return ff3d_launch_status();
}

// Rows per block: the height whose grid needs the least (rounds of the chip) x (time of a block ~ 2 + MT: a block streams all of
// its 256 x K weights whatever its height); ties -> the taller block.  FF3D_LINROWS_MT = 1..5 forces a height (A/B runs).
int rows_mt(int M, int N, int planes) {
  static const int forced = [] {
    const char* e = getenv("FF3D_LINROWS_MT");
    return e ? atoi(e) : 0;
  }();
  if (forced >= 1 && forced <= 5) return forced;
  static int cus[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!cus[dev & 63]) {
    hipDeviceProp_t prop;
    cus[dev & 63] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const long long slots = (long long)cus[dev & 63] * (planes == 1 ? 2 : 1);
  const long long n_tiles = (N + LR_BN - 1) / LR_BN;
  int best = 1;
  long long best_cost = -1;
  for (int mt = 1; mt <= 5; ++mt) {
    const long long blocks = ((M + 16 * mt - 1) / (16 * mt)) * n_tiles;
    const long long cost = ((blocks + slots - 1) / slots) * (2 + mt);
    if (best_cost < 0 || cost <= best_cost) best = mt, best_cost = cost;
  }
  return best;
}

template <int PLANES, bool LN>
int rows_dispatch(const RowsParams& p, hipStream_t s) {
  switch (rows_mt(p.M, p.N, PLANES)) {
    case 1: return launch_rows<1, PLANES, LN>(p, s);
    case 2: return launch_rows<2, PLANES, LN>(p, s);
    case 3: return launch_rows<3, PLANES, LN>(p, s);
    case 4: return launch_rows<4, PLANES, LN>(p, s);
    default: return launch_rows<5, PLANES, LN>(p, s);
  }
}

}  // namespace

extern "C" int ff3d_linear_rows(const float* a, const float* a2, int n_split, int64_t lda, const void* w_hi, const void* w_lo,
                                const int32_t* w_exp, const float* bias, int act, const float* residual, const float* gamma,
                                const float* beta, float eps, const float* pos, float* out, float* out_pos, int64_t ldc, int M,
                                int N, int K, ff3d_stream_t stream) {
  FF3D_REQUIRE(a && w_hi && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(M > 0 && N > 0 && K > 0 && K % LR_BK == 0 && lda >= K && ldc >= N && (act == 0 || act == 1), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE((long long)(N + 1) * K * 2 < (1ll << 32), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(a) && ff3d_aligned16(w_hi) && (!w_lo || ff3d_aligned16(w_lo)) && lda % 4 == 0, FF3D_ERR_ALIGNMENT);
  FF3D_REQUIRE(!a2 || (n_split > 0 && n_split < N && n_split % LR_BN == 0 && ff3d_aligned16(a2)), FF3D_ERR_BAD_SHAPE);
  const bool ln = residual != nullptr;
  if (ln) {
    FF3D_REQUIRE(gamma && beta && (!out_pos || pos), FF3D_ERR_NULL);
    FF3D_REQUIRE(N == LR_BN && ldc == N && act == 0 && !a2, FF3D_ERR_BAD_SHAPE);       // a block owns whole rows
    FF3D_REQUIRE(ff3d_aligned16(residual) && ff3d_aligned16(gamma) && ff3d_aligned16(beta) && ff3d_aligned16(out) &&
                     (!bias || ff3d_aligned16(bias)) && (!pos || ff3d_aligned16(pos)) && (!out_pos || ff3d_aligned16(out_pos)),
                 FF3D_ERR_ALIGNMENT);
  }
  RowsParams p{};
  p.a = a, p.a2 = a2, p.n_split = a2 ? n_split : 0;
  p.res = residual, p.gamma = gamma, p.beta = beta, p.pos = pos, p.out2 = out_pos, p.eps = eps;
  p.w_hi = static_cast<const _Float16*>(w_hi), p.w_lo = static_cast<const _Float16*>(w_lo), p.w_exp = w_exp;
  p.bias = bias, p.out = out, p.lda = lda, p.ldc = ldc, p.M = M, p.N = N, p.K = K, p.act = act;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (w_lo) return ln ? rows_dispatch<2, true>(p, s) : rows_dispatch<2, false>(p, s);
  return ln ? rows_dispatch<1, true>(p, s) : rows_dispatch<1, false>(p, s);
}
