// Query-side dense projections of the decoder (QKV / output projections of mmcv MultiheadAttention and
// MultiScaleDeformableAttention, FFN, positional MLPs, roi_mlp.1-2, the prediction heads' first layer; reached from
// FD:870-871, 914-922, 927-933, 939) on the fp16 matrix cores with fp32-class accuracy:
//     out (M, N) fp32 = act(A (M, K) fp32 @ W (N, K)^T + bias),   M = B * Nq (600 .. 32 000), K, N = 64 .. 1024.
// hipBLASLt runs these fp32 GEMMs on v_mfma_f32_16x16x1/x4 (157 TFLOP/s ceiling; 12 - 100 us each, 52 launches = 9.6 % of
// the batch-32 step and 15 % of the batch-4 step).  Here the arithmetic is splitmm.hip's - every operand a (hi, lo') fp16
// pair, three v_mfma_f32_16x16x32_f16 passes per product, fp32 accumulation - but the ACTIVATION operand arrives as plain
// fp32 and is split while it is staged, so no producer has to emit pairs and no exponent travels between kernels:
//   * range normalisation is PER ROW: pass 1 reads the block's row panel (L2-resident: the neighbouring column blocks and
//     pass 2 read it again) and takes every row's exponent e_r = floor(log2 max|x|) - 13; pass 2 scales by 2^-e_r (exact),
//     splits and writes the (hi, lo') K-step tiles to LDS; the epilogue multiplies by 2^(e_r + e_w).  Any fp32 magnitude
//     works, rows are independent (a huge query does not cost a small one its low bits);
//   * weights: the (hi, lo') planes + exponent of ops.split_weight_f16 (one zero row after row N - 1);
//   * the TRANSPOSED tile is accumulated (D^T = W A^T: the MFMA's A / B fragment layouts are symmetric), so a lane holds 4
//     consecutive output columns of one row: float4 bias loads and 16-byte stores;
//   * block = 256 threads = 4 waves, tile BM (64 | 32) rows x 128 columns, wave = all BM rows x 32 columns; K-step 32,
//     LDS double buffer (A tile written by ds_write_b128 after conversion, W tile by 16-byte LDS DMA), one barrier per step,
//     the next step's global loads in flight under the current step's 12 / 24 MFMAs per wave.  LDS rows are 64 B with the
//     chunk XOR-swizzle of splitmm.hip (conflict-free ds_read_b128 service groups).
#include "ff3d_common.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int LN_BN = 128, LN_BK = 32;

__device__ __forceinline__ int ln_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

struct LinearParams {
  const float* a;
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

template <int BM>
__global__ __launch_bounds__(256, 2) void linear_f16x3_kernel(LinearParams p) {
  constexpr int MT = BM / 16;                                        // 16-row tiles of the activation per wave
  constexpr int A_TILE = BM * LN_BK, B_TILE = LN_BN * LN_BK;          // halves per plane tile
  constexpr int BUF = 2 * A_TILE + 2 * B_TILE;
  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * BUF];      // [buf][A_hi | A_lo | W_hi | W_lo]
  __shared__ int s_exp[BM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int n_tiles = (p.N + LN_BN - 1) / LN_BN, m_tiles = (p.M + BM - 1) / BM;
  const unsigned lid = ff3d_xcd_remap(blockIdx.x, (unsigned)(n_tiles * m_tiles));
  const int n0 = (int)(lid % n_tiles) * LN_BN, m0 = (int)(lid / n_tiles) * BM;
  const int nk = p.K / LN_BK;

  // ---- activation staging geometry: BM rows x 4 chunks of 8 floats per K-step; BM = 64: one chunk per thread,
  //      BM = 32: threads 0..127 (waves 0, 1) stage, the others only move weights
  const bool a_thread = tid < BM * 4;
  const int a_row = (tid >> 2) % BM, a_q = tid & 3;
  const bool a_real = a_thread && (m0 + a_row < p.M);
  const float* a_ptr = p.a + (long long)min(m0 + a_row, p.M - 1) * p.lda + a_q * 8;

  // ---- pass 1: row exponents.  The 4 threads of a row scan its K floats (chunk a_q of every K-step), xor-shuffle max.
  float inv_scale = 1.f;
  if (a_thread) {
    float mx = 0.f;
    if (a_real) {
      for (int ks = 0; ks < nk; ++ks) {
        const float4 u = *reinterpret_cast<const float4*>(a_ptr + ks * LN_BK);
        const float4 v = *reinterpret_cast<const float4*>(a_ptr + ks * LN_BK + 4);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fmaxf(fabsf(u.z), fabsf(u.w))));
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    // max * 2^-e in [2^13, 2^14); zero rows keep e = 0; a NaN / inf row keeps its NaN / inf through the scaled split
    const int eb = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    const int e = (mx > 0.f) ? eb - 127 - 13 : 0;
    inv_scale = ff3d_pow2(-e);
    if (a_q == 0) s_exp[a_row] = e;
  }

  // ---- weight staging geometry: 128 rows x 4 chunks x 2 planes = 1024 16-byte pieces per K-step, 4 per thread (LDS DMA:
  //      lane-linear destination, swizzle applied on the per-lane source address)
  unsigned w_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int s = j * 256 + tid, row = s >> 2, n = n0 + row;
    w_off[j] = (unsigned)min(n, p.N) * (unsigned)p.K * 2u + (unsigned)(((s & 3) ^ ln_swz(row)) * 16);   // row N = the zero row
  }
  const _Float16 *w_hi = p.w_hi, *w_lo = p.w_lo;
  float4 ra0, ra1;
  auto load_a = [&](int ks) {
    if (a_real) {
      ra0 = *reinterpret_cast<const float4*>(a_ptr + ks * LN_BK);
      ra1 = *reinterpret_cast<const float4*>(a_ptr + ks * LN_BK + 4);
    } else {
      ra0 = ra1 = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto dma_w = [&](int ks, int buf) {
    _Float16* base = lds + buf * BUF + 2 * A_TILE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      _Float16* dst = base + (j * 256 + wave * 64) * 8;              // wave-uniform: 64 lanes x 16 B behind it
      ln_glds16(w_hi, w_off[j] + (unsigned)ks * 64u, dst);
      ln_glds16(w_lo, w_off[j] + (unsigned)ks * 64u, dst + B_TILE);
    }
  };
  auto store_a = [&](int buf) {
    if (!a_thread) return;
    const float f[8] = {ra0.x, ra0.y, ra0.z, ra0.w, ra1.x, ra1.y, ra1.z, ra1.w};
    half8 hh, ll;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = f[i] * inv_scale;
      const _Float16 h = (_Float16)x;
      hh[i] = h;
      ll[i] = (_Float16)((x - (float)h) * 2048.f);
    }
    const int o = a_row * 32 + ((a_q ^ ln_swz(a_row)) * 8);
    *reinterpret_cast<half8*>(lds + buf * BUF + o) = hh;
    *reinterpret_cast<half8*>(lds + buf * BUF + A_TILE + o) = ll;
  };

  f32x4 am[2][MT], ax[2][MT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m) am[t][m] = f32x4{0.f, 0.f, 0.f, 0.f}, ax[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_a(0);
  dma_w(0, 0);
  store_a(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this thread's DMA pieces + LDS stores have landed
  __syncthreads();

  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) {
      load_a(ks + 1);
      dma_w(ks + 1, buf ^ 1);
    }
    const _Float16* A = lds + buf * BUF;
    const _Float16* W = A + 2 * A_TILE;
    half8 wh[2], wl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = wave * 32 + t * 16 + fr;
      const int o = row * 32 + ((kq ^ ln_swz(row)) * 8);
      wh[t] = *reinterpret_cast<const half8*>(W + o);
      wl[t] = *reinterpret_cast<const half8*>(W + B_TILE + o);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int row = m * 16 + fr;
      const int o = row * 32 + ((kq ^ ln_swz(row)) * 8);
      const half8 ah = *reinterpret_cast<const half8*>(A + o);
      const half8 al = *reinterpret_cast<const half8*>(A + A_TILE + o);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        am[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], ah, am[t][m], 0, 0, 0);
        ax[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], al, ax[t][m], 0, 0, 0);
        ax[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], ah, ax[t][m], 0, 0, 0);
      }
    }
    if (ks + 1 < nk) store_a(buf ^ 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: lane (row fr of m-tile m, kq) holds columns n0 + wave*32 + t*16 + 4*kq .. +3 of row m0 + m*16 + fr
  const int we = ff3d_ld_exp(p.w_exp);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int r = m * 16 + fr, gm = m0 + r;
    if (gm >= p.M) continue;
    const float sc = ff3d_pow2(s_exp[r] + we);
    float* orow = p.out + (long long)gm * p.ldc;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = n0 + wave * 32 + t * 16 + 4 * kq;
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = (am[t][m][i] + ax[t][m][i] * (1.f / 2048.f)) * sc;
      if (n + 3 < p.N) {
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

}  // namespace

extern "C" int ff3d_linear_f16x3(const float* a, int64_t lda, const void* w_hi, const void* w_lo, const int32_t* w_exp,
                                 const float* bias, int act, float* out, int64_t ldc, int M, int N, int K,
                                 ff3d_stream_t stream) {
  FF3D_REQUIRE(a && w_hi && w_lo && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(M > 0 && N > 0 && K > 0 && K % LN_BK == 0 && lda >= K && ldc >= N && (act == 0 || act == 1), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE((long long)(N + 1) * K * 2 < (1ll << 32), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(a) && ff3d_aligned16(w_hi) && ff3d_aligned16(w_lo) && ff3d_aligned16(out) && lda % 4 == 0 &&
                   ldc % 4 == 0 && (!bias || ff3d_aligned16(bias)),
               FF3D_ERR_ALIGNMENT);
  LinearParams p{a, static_cast<const _Float16*>(w_hi), static_cast<const _Float16*>(w_lo), w_exp, bias, out, lda, ldc, M, N, K, act};
  const int n_tiles = (N + LN_BN - 1) / LN_BN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  ff3d_clear_error();
  // 64-row tiles once they fill the chip (two blocks per CU), 32-row tiles below
  if ((long long)((M + 63) / 64) * n_tiles >= 512)
    hipLaunchKernelGGL((linear_f16x3_kernel<64>), dim3((unsigned)(((M + 63) / 64) * n_tiles)), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((linear_f16x3_kernel<32>), dim3((unsigned)(((M + 31) / 32) * n_tiles)), dim3(256), 0, s, p);
  return ff3d_launch_status();
}
