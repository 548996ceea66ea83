// Weight gradient of a linear layer with fp32-class accuracy on the fp16 matrix cores of gfx950 (training path):
//     dW (N, K) = dY (M, N)^T  X (M, K),      db (N) = column sums of dY
// - the "TN" GEMM of every nn.Linear backward (FD:1166-1311 through the framework's autograd in the reference; hipBLASLt fp32 here
// until round 6: the ten M = 170 100 value_proj weight gradients of a 4-frame step ran at 58 TFLOP/s, 381 us each).
//
// Both operands are row-major fp32 with the REDUCTION index m as the slow dimension, while the MFMA wants, per lane, eight
// consecutive k of one row.  No transposing pass and no fp16 copy of the operands in HBM:
//   * a 512-thread block owns a 256 (n) x 128 (k) tile of dW and a slice of the rows (split-M: gridDim = tiles x S slices);
//     per step of 32 rows it DMAs the raw fp32 tiles dY[32][256] and X[32][128] into LDS (global_load_lds, 16 bytes per lane,
//     48 pieces of 1 KiB per step, double-buffered, one barrier per step);
//   * a wave (8 = 4 (n) x 2 (k), wave tile 64 x 64 = 4 x 4 MFMA tiles, two accumulators each) builds its fragments with
//     ds_read_b32: lane (fr = lane & 15, kq = lane >> 4) reads column fr of rows 4 t + kq, t = 0 .. 7 - ANY bijection between the 32
//     k-slots of v_mfma_f32_16x16x32_f16 and the 32 rows of the step is a valid reduction order as long as both operands use the
//     same one, and this one makes a 16-lane group read 16 consecutive dwords of one row; LDS rows are padded by 64 bytes so that
//     the two rows of a 32-lane service group sit 16 banks apart (no bank conflicts);
//   * the fp32 values are scaled by 2^-e (e from the tensor's measured maximum: |v| 2^-e < 2^14) and split in registers into
//     (hi, lo') = (fp16(v), fp16((v - hi) 2^11)); three MFMA passes per product (hi hi | hi lo' + lo' hi, second accumulator) as in
//     splitmm.hip; the conversion is repeated by the waves that share a fragment (2.7 x) - the VALU cost of that is the bound of
//     this kernel (~5 instructions per value), not the MFMA pipe: still 4 - 5 x the fp32-MFMA GEMM it replaces;
//   * partial tiles go to a (S, N, K) workspace, wgrad_reduce_kernel adds the slices IN ORDER (deterministic) and applies
//     2^(e_x + e_dy); the bias gradient rides along: the waves of k-tile 0 add up the raw dY values they convert anyway.
// Ragged shapes: rows beyond M and columns beyond N / K are fetched from a clamped (valid) address and zeroed at conversion.
#include <cstdlib>
#include <type_traits>

#include "ff3d_common.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int WG_BN = 256, WG_BK = 128, WG_BM = 32, WG_T = 512;
constexpr int WG_AROW = 1024 + 64;                 // bytes per dY row in LDS (256 floats + pad)
constexpr int WG_A_BYTES = WG_BM * WG_AROW;
constexpr int WG_BSLOT = 1024 + 64;                // bytes per X slot = two 128-float rows + pad
constexpr int WG_B_BYTES = (WG_BM / 2) * WG_BSLOT;
constexpr int WG_STAGE = WG_A_BYTES + WG_B_BYTES;  // 52 224
constexpr int WG_NSTAGE = 3;                      // DMA two steps ahead (one step ahead: 130 us at M = 170 100 - the round trip of a step's
                                                  // pieces from HBM is longer than a step)
constexpr int WG_LDS_BYTES = WG_NSTAGE * WG_STAGE;   // + 32 bytes of scratch behind the stages
constexpr int WG_NPART = 256;                      // partial maxima per tensor (ff3d_absmax_partials_f32)

struct WgradParams {
  const float *x, *dy;
  const float *amax_x, *amax_y;   // WG_NPART partial maxima of |x| / of |dy| each
  float* ws;                // (S, N, K) partial sums in units of 2^(e_x + e_dy)
  float* ws_b;              // (S, N) partial column sums of dY, or null
  int M, K, N, S, steps_per_slice;
  long long ldx, ldy;       // row strides in floats
};

// exponent e with max * 2^-e in [2^13, 2^14) (0 for an all-zero / non-finite tensor)
__device__ __forceinline__ int wg_exp_of(float mx) {
  const unsigned ex = (__float_as_uint(mx) >> 23) & 0xffu;
  if (ex == 0u || ex == 0xffu) return 0;
  return (int)ex - 127 - 13;
}

__global__ __launch_bounds__(1024) void absmax_partials_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
  __shared__ float red[16];
  float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
  const long long n4 = n >> 2, stride = (long long)gridDim.x * 1024;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  auto amax4 = [](float m, const float4& v) { return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w))); };
  long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {        // four independent 16-byte loads in flight per lane
    const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
    m0 = amax4(m0, a), m1 = amax4(m1, b), m2 = amax4(m2, c), m3 = amax4(m3, d);
  }
  for (; i < n4; i += stride) m0 = amax4(m0, x4[i]);
  float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x < 16) {
    m = red[threadIdx.x];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (threadIdx.x == 0) out[blockIdx.x] = m;
  }
}

// the two tensor maxima from their partials: every thread of a block of >= 256 threads returns both (LDS scratch: 2 x 4 floats)
__device__ __forceinline__ void wg_block_maxima(const float* amax_x, const float* amax_y, float* scratch, float& mx, float& my) {
  const int tid = threadIdx.x;
  float a = 0.f, b = 0.f;
  if (tid < WG_NPART) a = amax_x[tid], b = amax_y[tid];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o)), b = fmaxf(b, __shfl_xor(b, o));
  if (tid < WG_NPART && (tid & 63) == 0) scratch[tid >> 6] = a, scratch[4 + (tid >> 6)] = b;
  __syncthreads();
  mx = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
  my = fmaxf(fmaxf(scratch[4], scratch[5]), fmaxf(scratch[6], scratch[7]));
  __syncthreads();
}

__device__ __forceinline__ void wg_glds16(const float* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// eight fp32 values -> the (hi, lo') fragment pair
__device__ __forceinline__ void wg_split8(const float (&v)[8], float s, half8& hi, half8& lo) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float a = v[t] * s;
    const _Float16 h = (_Float16)a;
    hi[t] = h;
    lo[t] = (_Float16)((a - (float)h) * 2048.f);
  }
}

// ABL: timing ablations (experiments build only, WRONG results): 1 no MFMAs, 2 no conversion arithmetic, 4 no LDS fragment reads, 8 no DMA
template <int ABL = 0>
__global__ __launch_bounds__(WG_T, 1) void linear_wgrad_f16x3_kernel(WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* const scratch = reinterpret_cast<float*>(lds + WG_LDS_BYTES);   // (no static LDS in front of the DMA stages)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;
  const int n_tiles = (p.N + WG_BN - 1) / WG_BN, k_tiles = (p.K + WG_BK - 1) / WG_BK;
  // blocks go to the XCDs round-robin (block b -> XCD b % 8): the tiles of one row slice read the same dY rows (and, with several
  // n-tiles, the same X rows), so they sit on ONE XCD, adjacent in its dispatch order - the second reader hits that XCD's L2
  // (with fewer than 8 slices - many tiles, few rows - the plain order: tile fastest)
  const int tiles = n_tiles * k_tiles, xcd = blockIdx.x % FF3D_NUM_XCD, idx = blockIdx.x / FF3D_NUM_XCD;
  const bool by_xcd = p.S >= FF3D_NUM_XCD;
  const int tile = by_xcd ? idx % tiles : blockIdx.x % tiles;
  const int slice = by_xcd ? xcd + FF3D_NUM_XCD * (idx / tiles) : blockIdx.x / tiles;
  if (slice >= p.S) return;                              // (by_xcd: the grid is padded to a multiple of 8 slices)
  const int n0 = (tile / k_tiles) * WG_BN, k0 = (tile % k_tiles) * WG_BK;
  const int m0 = slice * p.steps_per_slice * WG_BM;
  const int m_end = min(p.M, m0 + p.steps_per_slice * WG_BM);
  const int steps = m_end > m0 ? (m_end - m0 + WG_BM - 1) / WG_BM : 0;

  float mx, my;
  wg_block_maxima(p.amax_x, p.amax_y, scratch, mx, my);
  const float sx = ff3d_pow2(-wg_exp_of(mx)), sy = ff3d_pow2(-wg_exp_of(my));

  // ---- DMA geometry.  dY: piece q = wave * 4 + a = row q of the step (256 floats = one 1 KiB piece).  X: piece j = wave * 2 + b =
  //      slot j: lanes 0-31 fetch row 4 (j >> 1) + (j & 1), lanes 32-63 the row two below (kq = 2, 3 of the same t).
  const int a_col = min(n0 + 4 * lane, p.N - 4);
  const int b_col = min(k0 + 4 * (lane & 31), p.K - 4);
  auto issue = [&](int st, int buf) {
    if (ABL & 8) return;
    const int mb = m0 + st * WG_BM;
    char* base = lds + buf * WG_STAGE;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int q = wave * 4 + a, m = min(mb + q, p.M - 1);
      wg_glds16(p.dy + (long long)m * p.ldy + a_col, base + q * WG_AROW);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int j = wave * 2 + b, row = 4 * (j >> 1) + (j & 1) + 2 * (lane >> 5), m = min(mb + row, p.M - 1);
      wg_glds16(p.x + (long long)m * p.ldx + b_col, base + WG_A_BYTES + j * WG_BSLOT);
    }
  };

  f32x4 acc_m[4][4], acc_x[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc_m[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}, acc_x[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_bias = p.ws_b && k0 == 0 && wk == 0;

  // fragment read offsets (bytes inside a stage): row 4 t + kq
  int a_rd[4], b_rd[4];
  bool a_ok[4], b_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ca = wn * 64 + i * 16 + fr, cb = wk * 64 + i * 16 + fr;
    a_rd[i] = kq * WG_AROW + ca * 4;
    b_rd[i] = WG_A_BYTES + (kq & 1) * WG_BSLOT + (kq >> 1) * 512 + cb * 4;
    a_ok[i] = n0 + ca < p.N;
    b_ok[i] = k0 + cb < p.K;
  }
  const bool ragged_cols = n0 + WG_BN > p.N || k0 + WG_BK > p.K;

  if (steps > 0) issue(0, 0);
  if (steps > 1) issue(1, 1);
  int stage = 0;
  for (int st = 0; st < steps; ++st) {
    if (st + 1 < steps)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // the 6 pieces of step st + 1 stay in flight
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                    // this step's tiles landed; every wave is done with the stage of step st - 1
    if (st + 2 < steps) issue(st + 2, stage >= 1 ? stage - 1 : WG_NSTAGE - 1);
    const char* sa = lds + stage * WG_STAGE;
    stage = stage + 1 == WG_NSTAGE ? 0 : stage + 1;
    const int rows_left = m_end - (m0 + st * WG_BM);              // rows of this step that exist
    const bool mask = ragged_cols || rows_left < WG_BM;
    half8 ah[4], al[4], bh[4], bl[4];
    // (two copies of the conversion: the masked one only runs in ragged tiles and in a slice's last step - merged into one, every
    //  value paid a compare + select: 120 of ~400 VALU instructions per step)
    auto convert = [&](auto masked) {
      constexpr bool MASKED = decltype(masked)::value;
      if (MASKED) asm volatile("; ragged step" ::: "memory");   // (keeps the optimiser from folding the two copies back into one)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = (ABL & 4) ? sy * (float)(t + st) : *reinterpret_cast<const float*>(sa + a_rd[i] + t * 4 * WG_AROW);
        if (MASKED) {
#pragma unroll
          for (int t = 0; t < 8; ++t) v[t] = (a_ok[i] && 4 * t + kq < rows_left) ? v[t] : 0.f;
        }
        if (want_bias) bsum[i] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        if (ABL & 2) {
#pragma unroll
          for (int t = 0; t < 8; ++t) ah[i][t] = al[i][t] = __builtin_bit_cast(_Float16, (unsigned short)__float_as_uint(v[t]));
        } else
          wg_split8(v, sy, ah[i], al[i]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = (ABL & 4) ? sx * (float)(t + st) : *reinterpret_cast<const float*>(sa + b_rd[j] + t * 2 * WG_BSLOT);
        if (MASKED) {
#pragma unroll
          for (int t = 0; t < 8; ++t) v[t] = (b_ok[j] && 4 * t + kq < rows_left) ? v[t] : 0.f;
        }
        if (ABL & 2) {
#pragma unroll
          for (int t = 0; t < 8; ++t) bh[j][t] = bl[j][t] = __builtin_bit_cast(_Float16, (unsigned short)__float_as_uint(v[t]));
        } else
          wg_split8(v, sx, bh[j], bl[j]);
      }
    };
    if (mask)
      convert(std::true_type{});
    else
      convert(std::false_type{});
    if (ABL & 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]), "v"(bh[i]), "v"(bl[i]));
      continue;
    }
    // pass-major order (convhalo.hip): dependent MFMAs are 16 instructions apart
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc_m[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc_m[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc_x[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc_x[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc_x[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc_x[i][j], 0, 0, 0);
  }

  // ---- partial tile: D row = 4 kq + r (n), column = fr (k)
  float* ws = p.ws + (long long)slice * p.N * p.K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + wk * 64 + j * 16 + fr;
      if (k >= p.K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + kq * 4 + r;
        if (n < p.N) ws[(long long)n * p.K + k] = acc_m[i][j][r] + acc_x[i][j][r] * (1.f / 2048.f);
      }
    }
  if (want_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = bsum[i];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const int n = n0 + wn * 64 + i * 16 + fr;
      if (kq == 0 && n < p.N) p.ws_b[(long long)slice * p.N + n] = s;
    }
  }
}

// dw[i] = 2^(e_x + e_dy) * sum over the slices, in a FIXED order: four groups of consecutive slices are added up by four threads
// (eight loads in flight each), then the four group sums in order.  Block = 64 outputs x 4 groups.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ ws_b,
                                                           const float* __restrict__ amax_x,
                                                           const float* __restrict__ amax_y, float* __restrict__ dw,
                                                           float* __restrict__ db, int S, long long nk, int N) {
  __shared__ float scratch[8];
  __shared__ float part[4][64];
  float mx, my;
  wg_block_maxima(amax_x, amax_y, scratch, mx, my);
  const float scale_x = ff3d_pow2(wg_exp_of(mx)), scale_y = ff3d_pow2(wg_exp_of(my));   // (two factors: the sum may leave the range)
  const int g = threadIdx.x >> 6, l = threadIdx.x & 63, per = (S + 3) / 4, k0 = g * per, k1 = min(S, k0 + per);
  auto group_sum = [&](const float* base, long long plane, long long i) {
    float s = 0.f;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(long long)(k + u) * plane + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < k1; ++k) s += base[(long long)k * plane + i];
    return s;
  };
  const long long i = (long long)blockIdx.x * 64 + l;
  part[g][l] = i < nk ? group_sum(ws, nk, i) : 0.f;
  __syncthreads();
  if (g == 0 && i < nk) dw[i] = (((part[0][l] + part[1][l]) + part[2][l]) + part[3][l]) * scale_x * scale_y;
  if (db && (long long)blockIdx.x * 64 < N) {          // (N <= nk: the first blocks also own the bias; block-uniform condition)
    __syncthreads();
    part[g][l] = i < N ? group_sum(ws_b, N, i) : 0.f;
    __syncthreads();
    if (g == 0 && i < N) db[i] = ((part[0][l] + part[1][l]) + part[2][l]) + part[3][l];
  }
}

}  // namespace

extern "C" int ff3d_absmax_partials_f32(const float* x, int64_t n, float* out256, ff3d_stream_t stream) {
  FF3D_REQUIRE(x && out256, FF3D_ERR_NULL);
  FF3D_REQUIRE(n > 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(x), FF3D_ERR_ALIGNMENT);
  ff3d_clear_error();
  hipLaunchKernelGGL(absmax_partials_kernel, dim3(WG_NPART), dim3(1024), 0, static_cast<hipStream_t>(stream), x, (long long)n,
                     out256);
  return ff3d_launch_status();
}

extern "C" int ff3d_linear_wgrad_slices(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  const int tiles = ((N + WG_BN - 1) / WG_BN) * ((K + WG_BK - 1) / WG_BK), steps = (M + WG_BM - 1) / WG_BM;
  int s = 256 / tiles;
  if (s < 1) s = 1;
  if (s > steps) s = steps;
  const int per = (steps + s - 1) / s;
  return (steps + per - 1) / per;                 // no empty slice
}

extern "C" int ff3d_linear_wgrad_f16x3(const float* x, int64_t ldx, const float* dy, int64_t ldy, const float* amax_x,
                                       const float* amax_dy, int M, int K, int N, float* dw, float* db, float* workspace,
                                       ff3d_stream_t stream) {
  FF3D_REQUIRE(x && dy && amax_x && amax_dy && dw && workspace, FF3D_ERR_NULL);
  FF3D_REQUIRE(M > 0 && K >= 4 && N >= 4 && K % 4 == 0 && N % 4 == 0 && ldx >= K && ldy >= N && ldx % 4 == 0 && ldy % 4 == 0,
               FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(x) && ff3d_aligned16(dy), FF3D_ERR_ALIGNMENT);
  hipStream_t s = static_cast<hipStream_t>(stream);
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_wgrad_f16x3_kernel<0>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS_BYTES + 32) == hipSuccess;
  FF3D_REQUIRE(attr_ok, FF3D_ERR_LAUNCH);
  const int S = ff3d_linear_wgrad_slices(M, K, N);
  const int tiles = ((N + WG_BN - 1) / WG_BN) * ((K + WG_BK - 1) / WG_BK), steps = (M + WG_BM - 1) / WG_BM;
  float* ws_b = db ? workspace + (long long)S * N * K : nullptr;
  WgradParams p{x, dy, amax_x, amax_dy, workspace, ws_b, M, K, N, S, (steps + S - 1) / S, (long long)ldx, (long long)ldy};
  ff3d_clear_error();
  const dim3 grid((unsigned)(tiles * (S >= 8 ? (S + 7) / 8 * 8 : S)));
#ifdef FF3D_BUILD_EXPERIMENTS
  static const int abl = [] {
    const char* e = getenv("FF3D_WG_ABLATE");
    return e ? atoi(e) : 0;
  }();
#define WG_ABL_CASE(n)                                                                                                        \
  case n:                                                                                                                     \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_wgrad_f16x3_kernel<n>),                                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS_BYTES + 32);                                 \
    hipLaunchKernelGGL(linear_wgrad_f16x3_kernel<n>, grid, dim3(WG_T), WG_LDS_BYTES + 32, s, p);                              \
    break;
  switch (abl) {
    WG_ABL_CASE(1) WG_ABL_CASE(2) WG_ABL_CASE(3) WG_ABL_CASE(4) WG_ABL_CASE(6) WG_ABL_CASE(7) WG_ABL_CASE(8) WG_ABL_CASE(9) WG_ABL_CASE(14) WG_ABL_CASE(15)
    default:
      hipLaunchKernelGGL(linear_wgrad_f16x3_kernel<0>, grid, dim3(WG_T), WG_LDS_BYTES + 32, s, p);
  }
#undef WG_ABL_CASE
#else
  hipLaunchKernelGGL(linear_wgrad_f16x3_kernel<0>, grid, dim3(WG_T), WG_LDS_BYTES + 32, s, p);
#endif
  const long long nk = (long long)N * K;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nk + 63) / 64)), dim3(256), 0, s, workspace, ws_b, amax_x, amax_dy, dw, db, S,
                     nk, N);
  return ff3d_launch_status();
}
