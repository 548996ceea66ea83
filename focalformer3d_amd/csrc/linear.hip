// Query-side dense projections of the decoder (QKV / output projections of mmcv MultiheadAttention and
// MultiScaleDeformableAttention, FFN, positional MLPs, roi_mlp.1-2, the prediction heads' first layer; reached from
// FD:870-871, 914-922, 927-933, 939) on the fp16 matrix cores with fp32-class accuracy:
//     out (M, N) fp32 = act(A (M, K) fp32 @ W (N, K)^T + bias),   M = B * Nq (600 .. 32 000), K, N = 64 .. 1024.
// hipBLASLt runs these fp32 GEMMs on v_mfma_f32_16x16x1/x4 (157 TFLOP/s ceiling; 12 - 100 us each, 52 launches = 9.6 % of
// the batch-32 step and 15 % of the batch-4 step).  Here the arithmetic is splitmm.hip's - every operand a (hi, lo') fp16
// pair, three v_mfma_f32_16x16x32_f16 passes per product, fp32 accumulation - but the ACTIVATION operand arrives as plain
// fp32 and is split while it is staged, so no producer has to emit pairs and no exponent travels between kernels:
//   * K is walked in super-chunks of 256.  A thread keeps its share of the block's (BM x 256) fp32 panel in REGISTERS (one
//     global read of the activation, all 16 / 8 float4 loads of a thread in flight at once), the row maximum over the chunk
//     is an in-register max + xor-shuffles, and the chunk is normalised by its own power of two per ROW: e = floor(log2
//     max|x|) - 13, scaling exact.  Any fp32 magnitude works, rows are independent (a huge query does not cost a small one its
//     low bits).  Each K-step's (hi, lo') tile is converted from the registers into a double-buffered LDS tile; a chunk's MFMA
//     accumulators are folded into the running fp32 sum with 2^(e_row + e_w) at the end of the chunk (once for K <= 256);
//     (first version: a separate pass over the panel for the row exponents, then per-step global loads one step ahead - every
//     K-step paid a global round trip: 40.8 us per launch at batch 32 against hipBLASLt's 49, slower than hipBLASLt at batch
//     <= 4, profiles/r03_c_*linear_v1*);
//   * weights: the (hi, lo') planes + exponent of ops.split_weight_f16 (one zero row after row N - 1), streamed by 16-byte
//     LDS DMA through a 3-stage ring TWO K-steps ahead (counted s_waitcnt vmcnt(4): the newest step stays in flight across
//     the barrier);
//   * the TRANSPOSED tile is accumulated (D^T = W A^T: the MFMA's A / B fragment layouts are symmetric), so a lane holds 4
//     consecutive output columns of one row: float4 bias loads and 16-byte stores;
//   * block = 256 threads = 4 waves, tile BM (64 | 32) rows x 128 columns, wave = all BM rows x 32 columns; one barrier per
//     K-step.  LDS rows are 64 B with the chunk XOR-swizzle of splitmm.hip (conflict-free ds_read_b128 service groups).
// Round 3, the launch count of the small-batch step (4 frames: ~155 launches, 50 of them these projections + 18 residual /
// LayerNorm kernels):
//   * ff3d_linear_dual_f16x3: the column tiles from n_split on read a SECOND activation (q | k from x + pos, v from x: the three
//     in-projections of nn.MultiheadAttention in one launch);
//   * ff3d_linear_add_ln_f16x3: a block owns whole output rows (NT = 4: 256 columns, wave = 64 columns) and the epilogue is the
//     decoder layer's post-norm step: LayerNorm(residual + A W^T + b) (+ the `+ query_pos` of the next operation as a second
//     output) - two-pass mean / variance over the row through shuffles and one LDS exchange between the four waves.
#include <cstdlib>

#include "ff3d_common.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int LN_BK = 32;

__device__ __forceinline__ int ln_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

struct LinearParams {
  const float *a, *a2;          // a2: activation of the column tiles n0 >= n_split (same lda), or null
  int n_split;
  long long a_kstep;            // K-sliced form (ff3d_linear_kslices_f16x3): column block n0 / n_split reads a + (n0 / n_split) * a_kstep
  const float *res, *gamma, *beta, *pos;   // LN variant: residual (M, N), LayerNorm affine, optional second output out2 = y + pos
  float* out2;
  float eps;
  const _Float16 *w_hi, *w_lo;
  const int* w_exp;
  const float* bias;
  float* out;
  long long lda, ldc;
  int M, N, K, act;
};

__device__ __forceinline__ void ln_glds16(const _Float16* base, unsigned byte_off, _Float16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(base) + byte_off,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Epilogue shared by the two kernels: lane (row fr of m-tile m, kq) holds columns n0 + wave*16*NT + t*16 + 4*kq .. +3 of row
// m0 + m*16 + fr in sum[t][m].
template <int BM, int NT, bool LN>
__device__ __forceinline__ void ln_epilogue(const LinearParams& p, f32x4 (&sum)[NT][BM / 16], float* s_red, int m0, int n0,
                                            int wave, int fr, int kq) {
  constexpr int LN_BN = 64 * NT, MT = BM / 16;
  if (LN) {
    // LayerNorm(residual + A W^T + b) over the N = 64 * NT columns this block owns (add_layer_norm_kernel's arithmetic:
    // mean, then the centred sum of squares), optional second output y + pos
    const float inv_n = 1.f / (float)LN_BN;
    float mean[MT], rstd[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int gm = min(m0 + m * 16 + fr, p.M - 1);
      float s1 = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n = wave * (16 * NT) + t * 16 + 4 * kq;
        const float4 b = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 r = *reinterpret_cast<const float4*>(p.res + (long long)gm * LN_BN + n);
        sum[t][m][0] += b.x + r.x, sum[t][m][1] += b.y + r.y, sum[t][m][2] += b.z + r.z, sum[t][m][3] += b.w + r.w;
        s1 += (sum[t][m][0] + sum[t][m][1]) + (sum[t][m][2] + sum[t][m][3]);
      }
      s1 += __shfl_xor(s1, 16);
      s1 += __shfl_xor(s1, 32);
      if (kq == 0) s_red[(m * 16 + fr) * 4 + wave] = s1;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4 r = *reinterpret_cast<const float4*>(s_red + (m * 16 + fr) * 4);
      mean[m] = ((r.x + r.y) + (r.z + r.w)) * inv_n;
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
      if (kq == 0) s_red[BM * 4 + (m * 16 + fr) * 4 + wave] = s2;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4 r = *reinterpret_cast<const float4*>(s_red + BM * 4 + (m * 16 + fr) * 4);
      rstd[m] = rsqrtf(((r.x + r.y) + (r.z + r.w)) * inv_n + p.eps);
      const int gm = m0 + m * 16 + fr;
      if (gm >= p.M) continue;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n = wave * (16 * NT) + t * 16 + 4 * kq;
        const float4 g = *reinterpret_cast<const float4*>(p.gamma + n), be = *reinterpret_cast<const float4*>(p.beta + n);
        float4 y;
        y.x = (sum[t][m][0] - mean[m]) * rstd[m] * g.x + be.x;
        y.y = (sum[t][m][1] - mean[m]) * rstd[m] * g.y + be.y;
        y.z = (sum[t][m][2] - mean[m]) * rstd[m] * g.z + be.z;
        y.w = (sum[t][m][3] - mean[m]) * rstd[m] * g.w + be.w;
        *reinterpret_cast<float4*>(p.out + (long long)gm * LN_BN + n) = y;
        if (p.out2) {
          const float4 q = *reinterpret_cast<const float4*>(p.pos + (long long)gm * LN_BN + n);
          *reinterpret_cast<float4*>(p.out2 + (long long)gm * LN_BN + n) = make_float4(y.x + q.x, y.y + q.y, y.z + q.z, y.w + q.w);
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
      const int n = n0 + wave * (16 * NT) + t * 16 + 4 * kq;
      if (n >= p.N) continue;
      float v[4] = {sum[t][m][0], sum[t][m][1], sum[t][m][2], sum[t][m][3]};
      if (vec && n + 3 < p.N) {
        if (p.bias) {
          const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += b.x, v[1] += b.y, v[2] += b.z, v[3] += b.w;
        }
        if (p.act)
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        *reinterpret_cast<float4*>(orow + n) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (n + i < p.N) {
            const float y = v[i] + (p.bias ? p.bias[n + i] : 0.f);
            orow[n + i] = p.act ? fmaxf(y, 0.f) : y;
          }
      }
    }
  }
}

template <int BM, int NT, bool LN>
__global__ __launch_bounds__(256, NT == 2 ? 2 : 1) void linear_f16x3_kernel(LinearParams p) {
  constexpr int LN_BN = 64 * NT, WJ = LN_BN / 64;                    // block columns; weight DMA pieces per thread and plane
  constexpr int MT = BM / 16;                                        // 16-row tiles of the activation per wave
  constexpr int TPR = 256 / BM;                                      // threads per activation row: 4 | 8
  constexpr int SPT = 32 / TPR;                                      // K-steps of a super-chunk a thread stages: 8 | 4
  constexpr int A_TILE = BM * LN_BK, B_TILE = LN_BN * LN_BK;          // halves per plane tile
  constexpr int A_BUF = 2 * A_TILE, W_STAGE = 2 * B_TILE;
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];      // [2][A_hi | A_lo]  [3][W_hi | W_lo]  int exp[2][BM]
  _Float16* const ldsW = lds + 2 * A_BUF;
  int* const s_exp = reinterpret_cast<int*>(ldsW + 3 * W_STAGE);
  float* const s_red = reinterpret_cast<float*>(s_exp + 2 * BM);     // LN: [2][BM][4 waves] partial sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int n_tiles = (p.N + LN_BN - 1) / LN_BN, m_tiles = (p.M + BM - 1) / BM;
  const unsigned lid = ff3d_xcd_remap(blockIdx.x, (unsigned)(n_tiles * m_tiles));
  const int n0 = (int)(lid % n_tiles) * LN_BN, m0 = (int)(lid / n_tiles) * BM;
  const int nk = p.K / LN_BK, nsc = (nk + 7) / 8;

  // ---- activation staging geometry: thread (row, chunk a_q, parity a_par) holds chunk a_q of the K-steps ks = j * (TPR / 4) +
  //      a_par of the current super-chunk (j < SPT): 8 floats = one 16-byte (hi) + one 16-byte (lo') LDS store per step
  const int a_row = tid / TPR, a_sub = tid % TPR, a_q = a_sub & 3, a_par = a_sub >> 2;
  const bool a_real = m0 + a_row < p.M;
  const float* a_ptr = (p.a_kstep ? p.a + (long long)(n0 / p.n_split) * p.a_kstep : ((p.a2 && n0 >= p.n_split) ? p.a2 : p.a)) +
                       (long long)min(m0 + a_row, p.M - 1) * p.lda + a_q * 8;
  const int a_lds = a_row * 32 + ((a_q ^ ln_swz(a_row)) * 8);
  float4 ra[2 * SPT];
  float inv_scale = 1.f;

  // ---- weight staging geometry: 128 rows x 4 chunks x 2 planes = 1024 16-byte pieces per K-step, 4 per thread (LDS DMA:
  //      lane-linear destination, swizzle applied on the per-lane source address)
  const _Float16 *w_hi = p.w_hi, *w_lo = p.w_lo;
  unsigned w_off[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int s = j * 256 + tid, row = s >> 2, n = n0 + row;
    w_off[j] = (unsigned)min(n, p.N) * (unsigned)p.K * 2u + (unsigned)(((s & 3) ^ ln_swz(row)) * 16);   // row N = the zero row
  }
  auto dma_w = [&](int g) {                                           // global K-step g -> ring stage g % 3
    _Float16* base = ldsW + (g % 3) * W_STAGE;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      _Float16* dst = base + (j * 256 + wave * 64) * 8;              // wave-uniform: 64 lanes x 16 B behind it
      ln_glds16(w_hi, w_off[j] + (unsigned)g * 64u, dst);
      ln_glds16(w_lo, w_off[j] + (unsigned)g * 64u, dst + B_TILE);
    }
  };
  // one super-chunk of the activation panel into registers + its row exponent (the address is clamped to a real row:
  // always load, select afterwards)
  auto load_chunk = [&](int sc) {
    const int steps = min(8, nk - sc * 8);
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      const int ks = j * (TPR / 4) + a_par;
      const bool on = a_real && ks < steps;
      const float* src = a_ptr + (long long)(sc * 8 + min(ks, steps - 1)) * LN_BK;
      const float4 t0 = *reinterpret_cast<const float4*>(src), t1 = *reinterpret_cast<const float4*>(src + 4);
      ra[2 * j] = make_float4(on ? t0.x : 0.f, on ? t0.y : 0.f, on ? t0.z : 0.f, on ? t0.w : 0.f);
      ra[2 * j + 1] = make_float4(on ? t1.x : 0.f, on ? t1.y : 0.f, on ? t1.z : 0.f, on ? t1.w : 0.f);
    }
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < 2 * SPT; ++j)
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(ra[j].x), fabsf(ra[j].y)), fmaxf(fabsf(ra[j].z), fabsf(ra[j].w))));
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    // max * 2^-e in [2^13, 2^14); zero rows keep e = 0; a NaN / inf row keeps its NaN / inf through the scaled split
    const int eb = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    const int e = (mx > 0.f) ? eb - 127 - 13 : 0;
    inv_scale = ff3d_pow2(-e);
    if (a_sub == 0) s_exp[(sc & 1) * BM + a_row] = e;
  };
  auto store_a = [&](int ks, int buf) {                               // K-step ks (0..7) of the chunk in registers -> A buffer
    if ((ks & (TPR / 4 - 1)) != a_par) return;
    half8 hh, ll;
#pragma unroll
    for (int j = 0; j < SPT; ++j)
      if (j == ks / (TPR / 4)) {
#define LN_SPLIT(i, val)                               \
  {                                                    \
    const float x_ = (val) * inv_scale;                \
    const _Float16 h_ = (_Float16)x_;                  \
    hh[i] = h_;                                        \
    ll[i] = (_Float16)((x_ - (float)h_) * 2048.f);     \
  }
        LN_SPLIT(0, ra[2 * j].x) LN_SPLIT(1, ra[2 * j].y) LN_SPLIT(2, ra[2 * j].z) LN_SPLIT(3, ra[2 * j].w)
        LN_SPLIT(4, ra[2 * j + 1].x) LN_SPLIT(5, ra[2 * j + 1].y) LN_SPLIT(6, ra[2 * j + 1].z) LN_SPLIT(7, ra[2 * j + 1].w)
#undef LN_SPLIT
      }
    *reinterpret_cast<half8*>(lds + buf * A_BUF + a_lds) = hh;
    *reinterpret_cast<half8*>(lds + buf * A_BUF + A_TILE + a_lds) = ll;
  };

  f32x4 sum[NT][MT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m) sum[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int we = ff3d_ld_exp(p.w_exp);

  dma_w(0);
  if (nk > 1) dma_w(1);
  for (int sc = 0; sc < nsc; ++sc) {
    const int steps = min(8, nk - sc * 8);
    load_chunk(sc);
    store_a(0, (sc * 8) & 1);
    // the first K-step's weights (issued two steps ago / in the prologue) and this thread's LDS stores have landed
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 am[NT][MT], ax[NT][MT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int m = 0; m < MT; ++m) am[t][m] = f32x4{0.f, 0.f, 0.f, 0.f}, ax[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks >= steps) break;                                          // block-uniform
      const int g = sc * 8 + ks;
      const bool ahead = g + 2 < nk;
      if (ahead) dma_w(g + 2);
      const _Float16* A = lds + (g & 1) * A_BUF;
      const _Float16* W = ldsW + (g % 3) * W_STAGE;
      half8 wh[NT], wl[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int row = wave * (16 * NT) + t * 16 + fr;
        const int o = row * 32 + ((kq ^ ln_swz(row)) * 8);
        wh[t] = *reinterpret_cast<const half8*>(W + o);
        wl[t] = *reinterpret_cast<const half8*>(W + B_TILE + o);
      }
      half8 ah[MT], al[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row = m * 16 + fr;
        const int o = row * 32 + ((kq ^ ln_swz(row)) * 8);
        ah[m] = *reinterpret_cast<const half8*>(A + o);
        al[m] = *reinterpret_cast<const half8*>(A + A_TILE + o);
      }
      // pass-major order (see convhalo.hip): the two dependent cross-term MFMAs of a tile are 2 * MT instructions apart
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
      if (ks + 1 < steps) {
        store_a(ks + 1, (g + 1) & 1);
        // the next step's weights have landed (the DMAs just issued for step g + 2 may stay in flight), LDS stores done
        if (ahead) {
          if (WJ == 2)
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
          else
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        } else
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    // fold the chunk into the running sum: 2^(row exponent of this chunk + weight exponent)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float sc_f = ff3d_pow2(s_exp[(sc & 1) * BM + m * 16 + fr] + we);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) sum[t][m][i] = fmaf(am[t][m][i] + ax[t][m][i] * (1.f / 2048.f), sc_f, sum[t][m][i]);
    }
    if (sc + 1 < nsc) __syncthreads();      // every wave is done with the A buffers / the exponent slot of two chunks ago
  }

  ln_epilogue<BM, NT, LN>(p, sum, s_red, m0, n0, wave, fr, kq);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the SMALL-M form (one to a few frames: 600 - 4 800 rows, <= 256 blocks, at most one per CU).  There the kernel above
// is a chain of latencies, not of work: a block's K loop is 8 - 32 barrier-separated steps whose weights are fetched only two
// steps ahead, so a step costs half a global-memory round trip (~1 us measured: 9 - 24 us per launch at 600 AND at 2 400 rows,
// profiles/r03_l_bench_b{1,4}_kernel_stats_last_step.txt), and every 256-wide super-chunk drains the queue and waits out its own
// activation load.  Same arithmetic, different data flow:
//   * the weight ring is NSTG deep (6 x 16 KiB at 128 columns, 4 x 32 KiB at 256) and runs NSTG - 1 steps ahead;
//   * a super-chunk of the activation is converted into LDS as a whole (all 8 K-steps' (hi, lo') tiles, 32 / 16 KiB) right
//     after it lands, which frees its registers for the NEXT super-chunk's loads - issued a full chunk ahead;
//   * those loads are inline-asm global_load_dwordx4 and every s_waitcnt vmcnt is counted by hand (weights and activations
//     share the in-order VM counter), so nothing ever drains the queue: no vmcnt(0) inside the K loop.
__device__ __forceinline__ void ln_wait_vm(int n) {   // s_waitcnt vmcnt(n), n a multiple of 4 (larger values clamp to 60: stricter)
  switch (n) {
#define FF3D_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    FF3D_VM(4) FF3D_VM(8) FF3D_VM(12) FF3D_VM(16) FF3D_VM(20) FF3D_VM(24) FF3D_VM(28) FF3D_VM(32) FF3D_VM(36) FF3D_VM(40)
    FF3D_VM(44) FF3D_VM(48) FF3D_VM(52) FF3D_VM(56)
#undef FF3D_VM
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
  }
}

template <int BM, int NT, bool LN, int NSTG>
__global__ __launch_bounds__(256, 1) void linear_small_f16x3_kernel(LinearParams p) {
  constexpr int LN_BN = 64 * NT, WJ = LN_BN / 64, PW = 2 * WJ, PF = NSTG - 1;   // PW: weight DMA pieces per thread and K-step
  constexpr int MT = BM / 16;
  constexpr int TPR = 256 / BM;                                      // threads per activation row: 8 | 16
  constexpr int SPT = 32 / TPR, NA = 2 * SPT;                        // K-steps a thread stages per super-chunk; its loads
  constexpr int A_TILE = BM * LN_BK, B_TILE = LN_BN * LN_BK;          // halves per plane tile
  constexpr int A_STEP = 2 * A_TILE, W_STAGE = 2 * B_TILE;
  static_assert(TPR >= 8 && NA % 4 == 0, "BM = 16 | 32");
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];      // [8 steps][A_hi | A_lo]  [NSTG][W_hi | W_lo]  exp  red
  _Float16* const ldsW = lds + 8 * A_STEP;
  int* const s_exp = reinterpret_cast<int*>(ldsW + NSTG * W_STAGE);
  float* const s_red = reinterpret_cast<float*>(s_exp + 2 * BM);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int n_tiles = (p.N + LN_BN - 1) / LN_BN, m_tiles = (p.M + BM - 1) / BM;
  const unsigned lid = ff3d_xcd_remap(blockIdx.x, (unsigned)(n_tiles * m_tiles));
  const int n0 = (int)(lid % n_tiles) * LN_BN, m0 = (int)(lid / n_tiles) * BM;
  const int nk = p.K / LN_BK, nsc = (nk + 7) / 8;

  const int a_row = tid / TPR, a_sub = tid % TPR, a_q = a_sub & 3, a_par = a_sub >> 2;
  const bool a_real = m0 + a_row < p.M;
  const float* a_ptr = (p.a_kstep ? p.a + (long long)(n0 / p.n_split) * p.a_kstep : ((p.a2 && n0 >= p.n_split) ? p.a2 : p.a)) +
                       (long long)min(m0 + a_row, p.M - 1) * p.lda + a_q * 8;
  const int a_lds = a_row * 32 + ((a_q ^ ln_swz(a_row)) * 8);
  f32x4 ra[NA];

  const _Float16 *w_hi = p.w_hi, *w_lo = p.w_lo;
  unsigned w_off[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int s = j * 256 + tid, row = s >> 2, n = n0 + row;
    w_off[j] = (unsigned)min(n, p.N) * (unsigned)p.K * 2u + (unsigned)(((s & 3) ^ ln_swz(row)) * 16);   // row N = the zero row
  }
  auto dma_w = [&](int g) {                                           // global K-step g -> ring stage g % NSTG
    _Float16* base = ldsW + (g % NSTG) * W_STAGE;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      _Float16* dst = base + (j * 256 + wave * 64) * 8;
      ln_glds16(w_hi, w_off[j] + (unsigned)g * 64u, dst);
      ln_glds16(w_lo, w_off[j] + (unsigned)g * 64u, dst + B_TILE);
    }
  };
  // NA loads of super-chunk sc (addresses clamped to real rows / K-steps; the values are selected after they land)
  auto issue_a = [&](int sc) {
    const int steps = min(8, nk - sc * 8);
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      const int ks = j * (TPR / 4) + a_par;
      const float* src = a_ptr + (long long)(sc * 8 + min(ks, steps - 1)) * LN_BK;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[2 * j]) : "v"(src) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(ra[2 * j + 1]) : "v"(src) : "memory");
    }
  };
  // the landed super-chunk: row exponent, all of this thread's K-steps converted into the LDS chunk image
  auto convert_a = [&](int sc) {
    const int steps = min(8, nk - sc * 8);
#pragma unroll
    for (int j = 0; j < NA; ++j) asm volatile("" : "+v"(ra[j]));     // uses stay behind the hand-counted wait
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      const bool on = a_real && j * (TPR / 4) + a_par < steps;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 v = ra[2 * j + h];
        v = f32x4{on ? v[0] : 0.f, on ? v[1] : 0.f, on ? v[2] : 0.f, on ? v[3] : 0.f};
        ra[2 * j + h] = v;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      }
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const int eb = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    const int e = (mx > 0.f) ? eb - 127 - 13 : 0;
    const float inv_scale = ff3d_pow2(-e);
    if (a_sub == 0) s_exp[(sc & 1) * BM + a_row] = e;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      const int ks = j * (TPR / 4) + a_par;
      half8 hh, ll;
#define LN_SPLIT(i, val)                               \
  {                                                    \
    const float x_ = (val) * inv_scale;                \
    const _Float16 h_ = (_Float16)x_;                  \
    hh[i] = h_;                                        \
    ll[i] = (_Float16)((x_ - (float)h_) * 2048.f);     \
  }
      LN_SPLIT(0, ra[2 * j][0]) LN_SPLIT(1, ra[2 * j][1]) LN_SPLIT(2, ra[2 * j][2]) LN_SPLIT(3, ra[2 * j][3])
      LN_SPLIT(4, ra[2 * j + 1][0]) LN_SPLIT(5, ra[2 * j + 1][1]) LN_SPLIT(6, ra[2 * j + 1][2]) LN_SPLIT(7, ra[2 * j + 1][3])
#undef LN_SPLIT
      *reinterpret_cast<half8*>(lds + ks * A_STEP + a_lds) = hh;
      *reinterpret_cast<half8*>(lds + ks * A_STEP + A_TILE + a_lds) = ll;
    }
  };

  f32x4 sum[NT][MT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m) sum[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int we = ff3d_ld_exp(p.w_exp);

  // VM queue of a thread, in order: A(0) | W(0) .. W(PF-1) | then per chunk sc: A(sc+1) at its start, W(g+PF) at step g
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (the address set-up loads are out of the counted window)
  issue_a(0);
#pragma unroll
  for (int q = 0; q < PF; ++q)
    if (q < nk) dma_w(q);
  for (int sc = 0; sc < nsc; ++sc) {
    const int g0 = sc * 8, steps = min(8, nk - g0);
    const bool more = sc + 1 < nsc;
    // A(sc) landed?  Issued behind it: the prologue's weights (sc = 0) or the weight groups of the previous chunk's steps
    const int behind = sc == 0 ? min(PF, nk) : max(0, min(8, nk - PF - (g0 - 8)));
    ln_wait_vm(behind * PW);
    convert_a(sc);
    if (more) issue_a(sc + 1);
    // W(g0) landed (behind it: W(g0+1 .. g0+PF-1) and A(sc+1)), the chunk image written
    ln_wait_vm((min(g0 + PF - 1, nk - 1) - g0) * PW + (more ? NA : 0));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // (raw: __syncthreads() would add its own vmcnt(0) and drain the weight ring)
    asm volatile("" ::: "memory");
    f32x4 am[NT][MT], ax[NT][MT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int m = 0; m < MT; ++m) am[t][m] = f32x4{0.f, 0.f, 0.f, 0.f}, ax[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks >= steps) break;                                          // block-uniform
      const int g = g0 + ks;
      if (g + PF < nk) dma_w(g + PF);                                  // into the stage step g - 1 read (barrier below)
      const _Float16* A = lds + ks * A_STEP;
      const _Float16* W = ldsW + (g % NSTG) * W_STAGE;
      half8 wh[NT], wl[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int row = wave * (16 * NT) + t * 16 + fr;
        const int o = row * 32 + ((kq ^ ln_swz(row)) * 8);
        wh[t] = *reinterpret_cast<const half8*>(W + o);
        wl[t] = *reinterpret_cast<const half8*>(W + B_TILE + o);
      }
      half8 ah[MT], al[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row = m * 16 + fr;
        const int o = row * 32 + ((kq ^ ln_swz(row)) * 8);
        ah[m] = *reinterpret_cast<const half8*>(A + o);
        al[m] = *reinterpret_cast<const half8*>(A + A_TILE + o);
      }
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
      if (ks + 1 < steps) {
        // W(g+1) landed: behind it W(g+2 .. g+PF) and, while it predates this chunk's start (ks + 1 < PF), A(sc+1)
        ln_wait_vm((min(g + PF, nk - 1) - (g + 1)) * PW + ((more && ks + 1 < PF) ? NA : 0));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // this step's fragment reads retired
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float sc_f = ff3d_pow2(s_exp[(sc & 1) * BM + m * 16 + fr] + we);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) sum[t][m][i] = fmaf(am[t][m][i] + ax[t][m][i] * (1.f / 2048.f), sc_f, sum[t][m][i]);
    }
    if (more) {                             // every wave is done with the chunk image and the last step's weight stage
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
  if (LN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ln_epilogue<BM, NT, LN>(p, sum, s_red, m0, n0, wave, fr, kq);
}

template <int BM, int NT, bool LN, int NSTG>
int launch_linear_small(const LinearParams& p, hipStream_t s) {
  constexpr int BN = 64 * NT;
  constexpr size_t lds_bytes = (size_t)(8 * 2 * BM * LN_BK + NSTG * 2 * BN * LN_BK) * sizeof(_Float16) + 2 * BM * sizeof(int) +
                               2 * BM * 4 * sizeof(float);
  static_assert(lds_bytes <= 160 * 1024, "LDS budget");
  static bool configured[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!configured[dev & 63]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_small_f16x3_kernel<BM, NT, LN, NSTG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
      return FF3D_ERR_LAUNCH;
    configured[dev & 63] = true;
  }
  const int blocks = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  ff3d_clear_error();
  hipLaunchKernelGGL((linear_small_f16x3_kernel<BM, NT, LN, NSTG>), dim3((unsigned)blocks), dim3(256), lds_bytes, s, p);
  return ff3d_launch_status();
}

template <int BM, int NT, bool LN>
int launch_linear(const LinearParams& p, hipStream_t s) {
  constexpr int BN = 64 * NT;
  constexpr size_t lds_bytes = (size_t)(2 * 2 * BM * LN_BK + 3 * 2 * BN * LN_BK) * sizeof(_Float16) + 2 * BM * sizeof(int) +
                               2 * BM * 4 * sizeof(float);
  static bool configured[64] = {};                // > 64 KiB of dynamic LDS has to be enabled once per kernel AND device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!configured[dev & 63]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_f16x3_kernel<BM, NT, LN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
      return FF3D_ERR_LAUNCH;
    configured[dev & 63] = true;
  }
  const int blocks = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  ff3d_clear_error();
  hipLaunchKernelGGL((linear_f16x3_kernel<BM, NT, LN>), dim3((unsigned)blocks), dim3(256), lds_bytes, s, p);
  return ff3d_launch_status();
}

int linear_checks(const float* a, int64_t lda, const void* w_hi, const void* w_lo, const float* out, int64_t ldc, int M, int N,
                  int K, int act) {
  FF3D_REQUIRE(a && w_hi && w_lo && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(M > 0 && N > 0 && K > 0 && K % LN_BK == 0 && lda >= K && ldc >= N && (act == 0 || act == 1), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE((long long)(N + 1) * K * 2 < (1ll << 32), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(a) && ff3d_aligned16(w_hi) && ff3d_aligned16(w_lo) && lda % 4 == 0, FF3D_ERR_ALIGNMENT);
  return FF3D_OK;
}

int linear_dispatch(const LinearParams& p, hipStream_t s) {
  // 64-row tiles once they fill the chip (two blocks per CU), 32-row tiles below; at most one block per CU: the small-M form
  // (FF3D_LIN_SMALL=0: never)
  static const bool small_on = [] {
    const char* e = getenv("FF3D_LIN_SMALL");
    return !(e && e[0] == '0');
  }();
  const int n_tiles = (p.N + 127) / 128;
  if (small_on && (long long)((p.M + 31) / 32) * n_tiles <= 256) return launch_linear_small<32, 2, false, 6>(p, s);
  // 64-row tiles once every CU has a block, 32-row tiles below.  (Measured, profiles/r03_w_linear_tile_height.txt: a rule that
  // picked 32-row tiles where they need fewer row-rounds of the 512 resident blocks - 19 200 x 256: 600 blocks of 64 rows = two
  // rounds, the second 17 % full - is SLOWER, 28.3 vs 26.6 us: a block streams all of W whatever its height; and at 9 600 rows
  // (300 blocks of 64) the 64-row tiles already win, 17.7 vs 20.4 us.)  FF3D_LIN_BM = 32 | 64 forces a height (A/B runs).
  static const int bm_force = [] {
    const char* e = getenv("FF3D_LIN_BM");
    return e ? atoi(e) : 0;
  }();
  const long long b64 = (long long)((p.M + 63) / 64) * n_tiles;
  if (bm_force == 64) return launch_linear<64, 2, false>(p, s);
  if (bm_force == 32) return launch_linear<32, 2, false>(p, s);
  return b64 >= 256 ? launch_linear<64, 2, false>(p, s) : launch_linear<32, 2, false>(p, s);
}

}  // namespace

extern "C" int ff3d_linear_f16x3(const float* a, int64_t lda, const void* w_hi, const void* w_lo, const int32_t* w_exp,
                                 const float* bias, int act, float* out, int64_t ldc, int M, int N, int K,
                                 ff3d_stream_t stream) {
  if (int st = linear_checks(a, lda, w_hi, w_lo, out, ldc, M, N, K, act)) return st;
  LinearParams p{};
  p.a = a, p.w_hi = static_cast<const _Float16*>(w_hi), p.w_lo = static_cast<const _Float16*>(w_lo), p.w_exp = w_exp;
  p.bias = bias, p.out = out, p.lda = lda, p.ldc = ldc, p.M = M, p.N = N, p.K = K, p.act = act;
  return linear_dispatch(p, static_cast<hipStream_t>(stream));
}

extern "C" int ff3d_linear_dual_f16x3(const float* a, const float* a2, int n_split, int64_t lda, const void* w_hi,
                                      const void* w_lo, const int32_t* w_exp, const float* bias, int act, float* out,
                                      int64_t ldc, int M, int N, int K, ff3d_stream_t stream) {
  if (int st = linear_checks(a, lda, w_hi, w_lo, out, ldc, M, N, K, act)) return st;
  FF3D_REQUIRE(a2, FF3D_ERR_NULL);
  FF3D_REQUIRE(n_split > 0 && n_split < N && n_split % 128 == 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(a2), FF3D_ERR_ALIGNMENT);
  LinearParams p{};
  p.a = a, p.a2 = a2, p.n_split = n_split;
  p.w_hi = static_cast<const _Float16*>(w_hi), p.w_lo = static_cast<const _Float16*>(w_lo), p.w_exp = w_exp;
  p.bias = bias, p.out = out, p.lda = lda, p.ldc = ldc, p.M = M, p.N = N, p.K = K, p.act = act;
  return linear_dispatch(p, static_cast<hipStream_t>(stream));
}

// Round 6 (third session): K SLICES as column blocks.  out[m, s * N + n] = sum_{k < K} a[m, s * K + k] * W'[s * N + n, k] for s < kslices:
// the planes hold W' = the slices of a (N, kslices * K) weight stacked along the rows, so the long-K projection of a few hundred rows
// (fc2 of the feed-forward step at 1 - 4 frames: 600 rows x K = 1024, where each of the 38 row-owning blocks of the fused
// [projection + LayerNorm] form streams the whole 1 MB weight: 23 us) becomes kslices x as many blocks that each stream one slice, and
// ff3d_sum_add_layer_norm adds the partial columns in slice order.  Per-row, per-256-chunk normalisation as in ff3d_linear_f16x3: the
// partial sums are exactly the chunk terms of the one-pass kernel.  No bias / activation here.
extern "C" int ff3d_linear_kslices_f16x3(const float* a, int64_t lda, int kslices, const void* w_hi, const void* w_lo,
                                         const int32_t* w_exp, float* out, int64_t ldc, int M, int N, int K, ff3d_stream_t stream) {
  FF3D_REQUIRE(kslices >= 2 && kslices <= 16 && N > 0 && N % 128 == 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(lda >= (int64_t)kslices * K && K % 4 == 0, FF3D_ERR_BAD_SHAPE);
  if (int st = linear_checks(a, lda, w_hi, w_lo, out, ldc, M, kslices * N, K, 0)) return st;
  LinearParams p{};
  p.a = a, p.n_split = N, p.a_kstep = K;
  p.w_hi = static_cast<const _Float16*>(w_hi), p.w_lo = static_cast<const _Float16*>(w_lo), p.w_exp = w_exp;
  p.out = out, p.lda = lda, p.ldc = ldc, p.M = M, p.N = kslices * N, p.K = K, p.act = 0;
  return linear_dispatch(p, static_cast<hipStream_t>(stream));
}

extern "C" int ff3d_linear_add_ln_f16x3(const float* a, int64_t lda, const void* w_hi, const void* w_lo, const int32_t* w_exp,
                                        const float* bias, const float* residual, const float* gamma, const float* beta,
                                        float eps, const float* pos, float* out, float* out_pos, int M, int N, int K,
                                        ff3d_stream_t stream) {
  if (int st = linear_checks(a, lda, w_hi, w_lo, out, N, M, N, K, 0)) return st;
  FF3D_REQUIRE(residual && gamma && beta && (!out_pos || pos), FF3D_ERR_NULL);
  FF3D_REQUIRE(N == 256, FF3D_ERR_BAD_SHAPE);                        // a block owns whole rows: 64 * NT columns
  FF3D_REQUIRE(ff3d_aligned16(residual) && ff3d_aligned16(gamma) && ff3d_aligned16(beta) && ff3d_aligned16(out) &&
                   (!bias || ff3d_aligned16(bias)) && (!pos || ff3d_aligned16(pos)) && (!out_pos || ff3d_aligned16(out_pos)),
               FF3D_ERR_ALIGNMENT);
  LinearParams p{};
  p.a = a, p.w_hi = static_cast<const _Float16*>(w_hi), p.w_lo = static_cast<const _Float16*>(w_lo), p.w_exp = w_exp;
  p.bias = bias, p.out = out, p.lda = lda, p.ldc = N, p.M = M, p.N = N, p.K = K;
  p.res = residual, p.gamma = gamma, p.beta = beta, p.pos = pos, p.out2 = out_pos, p.eps = eps;
  static const bool small_on = [] {
    const char* e = getenv("FF3D_LIN_SMALL");
    return !(e && e[0] == '0');
  }();
  if (small_on && (M + 15) / 16 <= 256) return launch_linear_small<16, 4, true, 4>(p, static_cast<hipStream_t>(stream));
  return launch_linear<32, 4, true>(p, static_cast<hipStream_t>(stream));
}
