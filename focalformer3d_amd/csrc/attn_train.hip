// Masked query self-attention for the TRAINING route of the decoder: forward and backward of the
// scaled-dot-product core of torch `nn.MultiheadAttention` inside mmcv `MultiheadAttention` (operation 'self_attn' of the
// decoder layer, reached from FD:927-933 with the attention masks of the ground-truth query groups, FD:849-858, and the
// configured attention dropout) - the step PyTorch runs on its fused SDPA kernels (AOTriton on ROCm).
//   out[b,i,h,:] = sum_j drop_ij * softmax_j(scale * <q_i, k_j> + mask_ij) * v_j
// q, k, v, out: element (b, n, h, d) at ptr + (b*N + n)*ld + h*Dh + d (column blocks of the in-projection GEMM's output);
// mask (B, N, N) uint8, non-zero = blocked (shared by the heads), or NULL; keep (B, heads, N, N) uint8 dropout keep-mask drawn
// by the framework's generator, or NULL; keep_scale = 1 / (1 - p).  fp32 throughout (N <= ~1000 queries, Dh 4 .. 64: the whole
// training step spends < 1 % here; no MFMA).
//   forward      8 lanes per query row (each takes every 8th key of a 64-key LDS tile), online softmax per lane, the 8 states
//                merged with xor-shuffles; writes out and lse_i = m_i + log l_i
//   backward dQ  8 lanes per query row: recompute p_ij from lse, D_i = <dO_i, O_i>, dQ_i = scale * sum_j dS_ij k_j
//   backward dKV 8 lanes per key: Q / dO / lse / D in 64-query LDS tiles, dV_j = sum_i p~_ij dO_i, dK_j = scale * sum_i dS_ij q_i
// Both backward kernels are plain loops + shuffles - no atomics, run-to-run identical.
#include <cstdlib>

#include "ff3d_common.h"

namespace {

struct MhaParams {
  const float *q, *k, *v, *out, *dout, *lse, *dsum;
  const uint8_t *mask, *keep;
  float *o, *lse_w, *dq, *dk, *dv, *dsum_w;
  long long ld_q, ld_k, ld_v, ld_o, ld_do, ld_dq, ld_dk, ld_dv;
  int N, heads;
  float scale, keep_scale;
};

constexpr int MT = 64;   // keys (forward, dQ) / queries (dK, dV) per LDS tile
constexpr int RB = 32;   // rows per block
constexpr int KS = 8;    // lanes per row: lane s of a row takes the tile entries jj = s, s + 8, ... (8 adjacent lanes, combined
                         // with xor-shuffles) - a row-per-thread first version took 0.44 - 0.58 ms per launch at 4 x 720 queries
constexpr int TB = RB * KS;

template <int DH>
__global__ __launch_bounds__(TB) void mha_fwd_kernel(MhaParams p) {
  __shared__ float sK[MT][DH + 1], sV[MT][DH + 1];
  const int tiles = (p.N + RB - 1) / RB;
  const int bh = blockIdx.x / tiles, t = blockIdx.x - bh * tiles, b = bh / p.heads, h = bh - b * p.heads;
  const int r = threadIdx.x / KS, sl = threadIdx.x % KS;
  const int i = t * RB + r;
  const bool live = i < p.N;
  const long long row0 = (long long)b * p.N;
  float q[DH], o[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) q[d] = live ? p.q[(row0 + i) * p.ld_q + h * DH + d] * p.scale : 0.f, o[d] = 0.f;
  float m = -INFINITY, l = 0.f;
  const uint8_t* mrow = (p.mask && live) ? p.mask + (row0 + i) * p.N : nullptr;
  const uint8_t* krow = (p.keep && live) ? p.keep + (((long long)b * p.heads + h) * p.N + i) * p.N : nullptr;
  for (int j0 = 0; j0 < p.N; j0 += MT) {
    __syncthreads();
    for (int e = threadIdx.x; e < MT * DH; e += TB) {
      const int jj = e / DH, d = e - jj * DH;
      const bool in = j0 + jj < p.N;
      sK[jj][d] = in ? p.k[(row0 + j0 + jj) * p.ld_k + h * DH + d] : 0.f;
      sV[jj][d] = in ? p.v[(row0 + j0 + jj) * p.ld_v + h * DH + d] : 0.f;
    }
    __syncthreads();
    if (!live) continue;
    const int nj = min(MT, p.N - j0);
    for (int jj = sl; jj < nj; jj += KS) {
      if (mrow && mrow[j0 + jj]) continue;
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) s = fmaf(q[d], sK[jj][d], s);
      if (s > m) {                                   // new running maximum: rescale what has been accumulated
        const float a = expf(m - s);
        l *= a;
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] *= a;
        m = s;
      }
      const float e = expf(s - m);
      l += e;
      const float w = krow ? (krow[j0 + jj] ? e * p.keep_scale : 0.f) : e;
#pragma unroll
      for (int d = 0; d < DH; ++d) o[d] = fmaf(w, sV[jj][d], o[d]);
    }
  }
  // combine the KS partial softmax states of the row (dead rows take part: whole waves shuffle)
#pragma unroll
  for (int x = 1; x < KS; x <<= 1) {
    const float m2 = __shfl_xor(m, x), l2 = __shfl_xor(l, x);
    const float mn = fmaxf(m, m2);
    const float a1 = (m == -INFINITY) ? 0.f : expf(m - mn), a2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = o[d] * a1 + __shfl_xor(o[d], x) * a2;
    m = mn;
  }
  if (live && sl == 0) {
    const float inv = 1.f / l;                       // a fully masked row gives 0 * inf = NaN, as torch's softmax of -inf does
#pragma unroll
    for (int d = 0; d < DH; ++d) p.o[(row0 + i) * p.ld_o + h * DH + d] = o[d] * inv;
    p.lse_w[((long long)b * p.heads + h) * p.N + i] = m + logf(l);
  }
}

template <int DH>
__global__ __launch_bounds__(TB) void mha_bwd_dq_kernel(MhaParams p) {
  __shared__ float sK[MT][DH + 1], sV[MT][DH + 1];
  const int tiles = (p.N + RB - 1) / RB;
  const int bh = blockIdx.x / tiles, t = blockIdx.x - bh * tiles, b = bh / p.heads, h = bh - b * p.heads;
  const int r = threadIdx.x / KS, sl = threadIdx.x % KS;
  const int i = t * RB + r;
  const bool live = i < p.N;
  const long long row0 = (long long)b * p.N;
  float q[DH], go[DH], dq[DH];
  float D = 0.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    q[d] = live ? p.q[(row0 + i) * p.ld_q + h * DH + d] * p.scale : 0.f;
    go[d] = live ? p.dout[(row0 + i) * p.ld_do + h * DH + d] : 0.f;
    D = fmaf(go[d], live ? p.out[(row0 + i) * p.ld_o + h * DH + d] : 0.f, D);
    dq[d] = 0.f;
  }
  const float lse = live ? p.lse[((long long)b * p.heads + h) * p.N + i] : 0.f;
  if (live && sl == 0) p.dsum_w[((long long)b * p.heads + h) * p.N + i] = D;
  const uint8_t* mrow = (p.mask && live) ? p.mask + (row0 + i) * p.N : nullptr;
  const uint8_t* krow = (p.keep && live) ? p.keep + (((long long)b * p.heads + h) * p.N + i) * p.N : nullptr;
  for (int j0 = 0; j0 < p.N; j0 += MT) {
    __syncthreads();
    for (int e = threadIdx.x; e < MT * DH; e += TB) {
      const int jj = e / DH, d = e - jj * DH;
      const bool in = j0 + jj < p.N;
      sK[jj][d] = in ? p.k[(row0 + j0 + jj) * p.ld_k + h * DH + d] : 0.f;
      sV[jj][d] = in ? p.v[(row0 + j0 + jj) * p.ld_v + h * DH + d] : 0.f;
    }
    __syncthreads();
    if (!live) continue;
    const int nj = min(MT, p.N - j0);
    for (int jj = sl; jj < nj; jj += KS) {
      if (mrow && mrow[j0 + jj]) continue;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) s = fmaf(q[d], sK[jj][d], s), dp = fmaf(go[d], sV[jj][d], dp);
      const float pr = expf(s - lse);
      if (krow) dp = krow[j0 + jj] ? dp * p.keep_scale : 0.f;
      const float ds = pr * (dp - D);
#pragma unroll
      for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, sK[jj][d], dq[d]);
    }
  }
#pragma unroll
  for (int x = 1; x < KS; x <<= 1)
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] += __shfl_xor(dq[d], x);
  if (live && sl == 0)
#pragma unroll
    for (int d = 0; d < DH; ++d) p.dq[(row0 + i) * p.ld_dq + h * DH + d] = dq[d] * p.scale;
}

template <int DH>
__global__ __launch_bounds__(TB) void mha_bwd_dkv_kernel(MhaParams p) {
  __shared__ float sQ[MT][DH + 1], sG[MT][DH + 1], sL[MT], sD[MT];
  const int tiles = (p.N + RB - 1) / RB;
  const int bh = blockIdx.x / tiles, t = blockIdx.x - bh * tiles, b = bh / p.heads, h = bh - b * p.heads;
  const int r = threadIdx.x / KS, sl = threadIdx.x % KS;
  const int j = t * RB + r;
  const bool live = j < p.N;
  const long long row0 = (long long)b * p.N, hrow = ((long long)b * p.heads + h) * p.N;
  float k[DH], v[DH], dk[DH], dv[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    k[d] = live ? p.k[(row0 + j) * p.ld_k + h * DH + d] : 0.f;
    v[d] = live ? p.v[(row0 + j) * p.ld_v + h * DH + d] : 0.f;
    dk[d] = dv[d] = 0.f;
  }
  for (int i0 = 0; i0 < p.N; i0 += MT) {
    __syncthreads();
    for (int e = threadIdx.x; e < MT * DH; e += TB) {
      const int ii = e / DH, d = e - ii * DH;
      const bool in = i0 + ii < p.N;
      sQ[ii][d] = in ? p.q[(row0 + i0 + ii) * p.ld_q + h * DH + d] * p.scale : 0.f;
      sG[ii][d] = in ? p.dout[(row0 + i0 + ii) * p.ld_do + h * DH + d] : 0.f;
    }
    if (threadIdx.x < MT && i0 + (int)threadIdx.x < p.N) {
      sL[threadIdx.x] = p.lse[hrow + i0 + threadIdx.x];
      sD[threadIdx.x] = p.dsum[hrow + i0 + threadIdx.x];
    }
    __syncthreads();
    if (!live) continue;
    const int ni = min(MT, p.N - i0);
    for (int ii = sl; ii < ni; ii += KS) {
      const int i = i0 + ii;
      if (p.mask && p.mask[(row0 + i) * p.N + j]) continue;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) s = fmaf(sQ[ii][d], k[d], s), dp = fmaf(sG[ii][d], v[d], dp);
      const float pr = expf(s - sL[ii]);
      float w = pr;
      if (p.keep) {
        const bool kp = p.keep[(hrow + i) * p.N + j] != 0;
        w = kp ? pr * p.keep_scale : 0.f;
        dp = kp ? dp * p.keep_scale : 0.f;
      }
      const float ds = pr * (dp - sD[ii]);
#pragma unroll
      for (int d = 0; d < DH; ++d) dv[d] = fmaf(w, sG[ii][d], dv[d]), dk[d] = fmaf(ds, sQ[ii][d], dk[d]);   // (sQ carries the scale)
    }
  }
#pragma unroll
  for (int x = 1; x < KS; x <<= 1)
#pragma unroll
    for (int d = 0; d < DH; ++d) dk[d] += __shfl_xor(dk[d], x), dv[d] += __shfl_xor(dv[d], x);
  if (live && sl == 0)
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      p.dk[(row0 + j) * p.ld_dk + h * DH + d] = dk[d];
      p.dv[(row0 + j) * p.ld_dv + h * DH + d] = dv[d];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the same three kernels on the matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate - bitwise an
// fmaf chain), in the layout of attn.hip's inference kernel: a 256-thread block owns 64 rows of one (frame, head), a wave 16 of
// them; the other side arrives in 64-row tiles through LDS.  Scores are computed TRANSPOSED, so a lane holds 4 rows of the
// staged side for one of its own rows (lane j = lane & 15 <-> own row j, g = lane >> 4 <-> staged rows 4 g + r): the exponentials
// are directly the B operand of the accumulating MFMAs, nothing goes through LDS a second time.
//   two LDS layouts of a staged 64 x DH tile X:
//     "score" layout  [16-row tile][dim / 2][16 rows][2]: A operand of  S^T = X Y^T  (A[i = row][k = dim])
//     "value" layout  row-major, row stride DH + 4:        A operand of  Z^T += X^T W  (A[i = dim][k = row])
//   forward   staged = keys:    S^T = K Q^T,  P = exp(S^T - m),  O^T += V^T P~       (K score layout, V value layout)
//   dQ        staged = keys:    S^T = K Q^T,  dP^T = V dO^T,  dS = P (dP~ - D),  dQ^T += K^T dS     (K both layouts, V score layout)
//   dK, dV    staged = queries: S = Q K^T,  dP = dO V^T,  dV^T += dO^T P~,  dK^T += Q^T dS          (Q, dO both layouts)
// Scalar kernels above: 164 / 162 / 292 us per layer at 4 x 720 queries, 8 heads of 32 (1 LDS read per FMA); kept for head sizes
// that are not multiples of 16 and unaligned operands.  FF3D_MHA_TRAIN_SCALAR=1 forces them (A/B runs).
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int XT = 64;

// A staged 64 x DH tile is loaded into registers one tile AHEAD (float4 pieces: piece e = tid + 256 i is dims 4 u .. 4 u + 3 of row kk)
// and written to LDS - in either layout, from the same registers - after the barrier that retires the previous tile: the global
// round trip runs under the previous tile's MFMAs (the first MFMA version loaded and stored in one step: forward 77 - 83 us).
template <int DH>
struct RowRegs {
  float4 v[XT * DH / 4 / 256];
};
template <int DH>
__device__ __forceinline__ void load_rows(RowRegs<DH>& rr, const float* x, long long ld, int r0, int N) {
#pragma unroll
  for (int i = 0; i < XT * DH / 4 / 256; ++i) {
    const int e = threadIdx.x + 256 * i, kk = e / (DH / 4), u = e - kk * (DH / 4);
    rr.v[i] = r0 + kk < N ? *reinterpret_cast<const float4*>(x + (long long)(r0 + kk) * ld + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int DH>
__device__ __forceinline__ void store_score(float* dst, const RowRegs<DH>& rr, float scale) {
#pragma unroll
  for (int i = 0; i < XT * DH / 4 / 256; ++i) {
    const int e = threadIdx.x + 256 * i, kk = e / (DH / 4), u = e - kk * (DH / 4);
    float* d = &dst[(kk >> 4) * 16 * DH + (2 * u) * 32 + (kk & 15) * 2];          // float2 pieces 2 u and 2 u + 1 of the row
    *reinterpret_cast<float2*>(d) = make_float2(rr.v[i].x * scale, rr.v[i].y * scale);
    *reinterpret_cast<float2*>(d + 32) = make_float2(rr.v[i].z * scale, rr.v[i].w * scale);
  }
}
template <int DH>
__device__ __forceinline__ void store_value(float* dst, const RowRegs<DH>& rr, float scale) {
  constexpr int VS = DH + 4;
#pragma unroll
  for (int i = 0; i < XT * DH / 4 / 256; ++i) {
    const int e = threadIdx.x + 256 * i, kk = e / (DH / 4), u = e - kk * (DH / 4);
    *reinterpret_cast<float4*>(&dst[kk * VS + 4 * u]) = make_float4(rr.v[i].x * scale, rr.v[i].y * scale, rr.v[i].z * scale, rr.v[i].w * scale);
  }
}
// S^T tile (16 staged rows x 16 own rows) from a score-layout tile and the own rows' B-operand registers
template <int DH>
__device__ __forceinline__ f32x4 score_tile(const float* tile, const float (&breg)[DH / 4], int j, int g) {
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < DH / 4; ++c) {
    const int dim = 4 * c + g;
    s = __builtin_amdgcn_mfma_f32_16x16x4f32(tile[(dim >> 1) * 32 + j * 2 + (dim & 1)], breg[c], s, 0, 0, 0);
  }
  return s;
}
// Z^T (DH dims x 16 own rows) += X^T W for the 16 staged rows of a value-layout tile; w[r] belongs to staged row 4 g + r
template <int DH>
__device__ __forceinline__ void accum_tile(f32x4 (&z)[DH / 16], const float* vt, const float (&w)[4], int j, int g) {
  constexpr int VS = DH + 4;
#pragma unroll
  for (int d = 0; d < DH / 16; ++d)
#pragma unroll
    for (int r = 0; r < 4; ++r) z[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[(4 * g + r) * VS + d * 16 + j], w[r], z[d], 0, 0, 0);
}
// the own rows' B operand: B[k = g][j] = x[row][4 c + g]
template <int DH>
__device__ __forceinline__ void load_breg(float (&b)[DH / 4], const float* row, int g, float scale) {
#pragma unroll
  for (int c = 0; c < DH / 4; ++c) b[c] = row[4 * c + g] * scale;
}

// The 64 x 64 byte tile rows [r0, r0 + 64) x columns [c0, c0 + 64) of a (.., N, N) uint8 matrix (mask: (B, N, N); keep: (B, heads, N, N))
// in LDS, row stride 17 words; entries outside the matrix = `fill`.  One 16-byte piece per thread (a global load per piece when N and
// the base allow it): the kernels read their bytes from LDS - a byte load from global memory per score, consumed at once, cost the
// first MFMA version more than the MFMAs saved (forward 141 us against 164 scalar).
constexpr int BW = 17;
struct ByteRegs {
  uint32_t w[4];
};
__device__ __forceinline__ void load_bytes(ByteRegs& br, const uint8_t* mat, int r0, int c0, int N, uint8_t fill) {
  const int row = threadIdx.x >> 2, chunk = threadIdx.x & 3, i = r0 + row, c = c0 + 16 * chunk;
  const uint32_t f4 = 0x01010101u * fill;
  br.w[0] = br.w[1] = br.w[2] = br.w[3] = f4;
  if (i < N && c < N) {
    const uint8_t* src = mat + (long long)i * N + c;
    if ((N & 15) == 0 && (reinterpret_cast<uintptr_t>(mat) & 15u) == 0) {
      const uint4 v = *reinterpret_cast<const uint4*>(src);
      br.w[0] = v.x, br.w[1] = v.y, br.w[2] = v.z, br.w[3] = v.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t x = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) x |= (uint32_t)(c + 4 * q + e < N ? src[4 * q + e] : fill) << (8 * e);
        br.w[q] = x;
      }
    }
  }
}
__device__ __forceinline__ void store_bytes(uint32_t* dst, const ByteRegs& br) {
  const int row = threadIdx.x >> 2, chunk = threadIdx.x & 3;
#pragma unroll
  for (int q = 0; q < 4; ++q) dst[row * BW + chunk * 4 + q] = br.w[q];
}

template <int DH>
__global__ __launch_bounds__(256) void mha_fwd_mfma_kernel(MhaParams p) {
  constexpr int VS = DH + 4;
  __shared__ __attribute__((aligned(16))) float sK[XT * DH], sV[XT * VS];
  __shared__ uint32_t sM[XT * BW], sP[XT * BW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int tiles = (p.N + XT - 1) / XT;
  const int bh = blockIdx.x / tiles, qt = blockIdx.x - bh * tiles, b = bh / p.heads, h = bh - b * p.heads;
  const int q0 = qt * XT + wave * 16, qi = min(q0 + j, p.N - 1);
  const long long row0 = (long long)b * p.N, hrow = ((long long)b * p.heads + h) * p.N;
  float qreg[DH / 4];
  load_breg<DH>(qreg, p.q + (row0 + qi) * p.ld_q + h * DH, g, p.scale);
  f32x4 o[DH / 16];
#pragma unroll
  for (int d = 0; d < DH / 16; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const bool has_mask = p.mask != nullptr, has_keep = p.keep != nullptr;
  RowRegs<DH> rk, rv;
  ByteRegs bm, bp;
  auto fetch = [&](int k0) {
    load_rows<DH>(rk, p.k + row0 * p.ld_k + h * DH, p.ld_k, k0, p.N);
    load_rows<DH>(rv, p.v + row0 * p.ld_v + h * DH, p.ld_v, k0, p.N);
    if (has_mask) load_bytes(bm, p.mask + row0 * p.N, qt * XT, k0, p.N, 1);
    if (has_keep) load_bytes(bp, p.keep + hrow * p.N, qt * XT, k0, p.N, 0);
  };
  fetch(0);
  for (int k0 = 0; k0 < p.N; k0 += XT) {
    __syncthreads();
    store_score<DH>(sK, rk, 1.f);
    store_value<DH>(sV, rv, 1.f);
    if (has_mask) store_bytes(sM, bm);
    if (has_keep) store_bytes(sP, bp);
    __syncthreads();
    if (k0 + XT < p.N) fetch(k0 + XT);
#pragma unroll
    for (int t = 0; t < XT / 16; ++t) {
      if (k0 + t * 16 >= p.N) break;
      const f32x4 s0 = score_tile<DH>(&sK[t * 16 * DH], qreg, j, g);
      const int kb = k0 + t * 16 + 4 * g;
      const uint32_t mw = has_mask ? sM[(wave * 16 + j) * BW + t * 4 + g] : 0u;        // bytes r = keys kb + r of this lane's query
      const uint32_t kw = has_keep ? sP[(wave * 16 + j) * BW + t * 4 + g] : 0xffffffffu;
      float s[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] = (kb + r >= p.N || ((mw >> (8 * r)) & 0xffu)) ? -INFINITY : s0[r];
      float m_loc = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
      m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 16));
      m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 32));
      const float m_new = fmaxf(m_run, m_loc);
      const float m_use = m_new == -INFINITY ? 0.f : m_new;      // (every key so far blocked: keep the state empty, no NaN)
      const float alpha = expf(m_run - m_use);
      float pr[4], l_loc = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[r] = expf(s[r] - m_use);
        l_loc += pr[r];
        if (has_keep) pr[r] = ((kw >> (8 * r)) & 0xffu) ? pr[r] * p.keep_scale : 0.f;
      }
      l_loc += __shfl_xor(l_loc, 16);
      l_loc += __shfl_xor(l_loc, 32);
      l_run = l_run * alpha + l_loc;
      m_run = m_new;
#pragma unroll
      for (int d = 0; d < DH / 16; ++d) o[d][0] *= alpha, o[d][1] *= alpha, o[d][2] *= alpha, o[d][3] *= alpha;
      accum_tile<DH>(o, &sV[t * 16 * VS], pr, j, g);
    }
  }
  if (q0 + j < p.N) {
    const float inv = 1.f / l_run;                     // a fully blocked row: 0 * inf = NaN, as torch's softmax of -inf
    float* op = p.o + (row0 + q0 + j) * p.ld_o + h * DH + 4 * g;
#pragma unroll
    for (int d = 0; d < DH / 16; ++d)
      *reinterpret_cast<float4*>(op + d * 16) = make_float4(o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv);
    if (g == 0) p.lse_w[hrow + q0 + j] = m_run + logf(l_run);
  }
}

template <int DH>
__global__ __launch_bounds__(256) void mha_bwd_dq_mfma_kernel(MhaParams p) {
  constexpr int VS = DH + 4;
  __shared__ __attribute__((aligned(16))) float sK[XT * DH], sV[XT * DH], sKv[XT * VS];
  __shared__ uint32_t sM[XT * BW], sP[XT * BW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int tiles = (p.N + XT - 1) / XT;
  const int bh = blockIdx.x / tiles, qt = blockIdx.x - bh * tiles, b = bh / p.heads, h = bh - b * p.heads;
  const int q0 = qt * XT + wave * 16, qi = min(q0 + j, p.N - 1);
  const long long row0 = (long long)b * p.N, hrow = ((long long)b * p.heads + h) * p.N;
  float qreg[DH / 4], greg[DH / 4];
  load_breg<DH>(qreg, p.q + (row0 + qi) * p.ld_q + h * DH, g, p.scale);
  load_breg<DH>(greg, p.dout + (row0 + qi) * p.ld_do + h * DH, g, 1.f);
  float D = 0.f;
  {
    const float* orow = p.out + (row0 + qi) * p.ld_o + h * DH;
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) D = fmaf(greg[c], orow[4 * c + g], D);
    D += __shfl_xor(D, 16);
    D += __shfl_xor(D, 32);
  }
  const float lse = p.lse[hrow + qi];
  if (g == 0 && q0 + j < p.N) p.dsum_w[hrow + q0 + j] = D;
  f32x4 dq[DH / 16];
#pragma unroll
  for (int d = 0; d < DH / 16; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool has_mask = p.mask != nullptr, has_keep = p.keep != nullptr;
  RowRegs<DH> rk, rv;
  ByteRegs bm, bp;
  auto fetch = [&](int k0) {
    load_rows<DH>(rk, p.k + row0 * p.ld_k + h * DH, p.ld_k, k0, p.N);
    load_rows<DH>(rv, p.v + row0 * p.ld_v + h * DH, p.ld_v, k0, p.N);
    if (has_mask) load_bytes(bm, p.mask + row0 * p.N, qt * XT, k0, p.N, 1);
    if (has_keep) load_bytes(bp, p.keep + hrow * p.N, qt * XT, k0, p.N, 0);
  };
  fetch(0);
  for (int k0 = 0; k0 < p.N; k0 += XT) {
    __syncthreads();
    store_score<DH>(sK, rk, 1.f);
    store_score<DH>(sV, rv, 1.f);
    store_value<DH>(sKv, rk, 1.f);
    if (has_mask) store_bytes(sM, bm);
    if (has_keep) store_bytes(sP, bp);
    __syncthreads();
    if (k0 + XT < p.N) fetch(k0 + XT);
#pragma unroll
    for (int t = 0; t < XT / 16; ++t) {
      if (k0 + t * 16 >= p.N) break;
      const f32x4 s = score_tile<DH>(&sK[t * 16 * DH], qreg, j, g);
      const f32x4 dp = score_tile<DH>(&sV[t * 16 * DH], greg, j, g);
      const int kb = k0 + t * 16 + 4 * g;
      const uint32_t mw = has_mask ? sM[(wave * 16 + j) * BW + t * 4 + g] : 0u;
      const uint32_t kw = has_keep ? sP[(wave * 16 + j) * BW + t * 4 + g] : 0xffffffffu;
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = kb + r < p.N && !((mw >> (8 * r)) & 0xffu);
        const float pr = ok ? expf(s[r] - lse) : 0.f;
        const float d_ = ((kw >> (8 * r)) & 0xffu) ? (has_keep ? dp[r] * p.keep_scale : dp[r]) : 0.f;
        ds[r] = pr * (d_ - D);
      }
      accum_tile<DH>(dq, &sKv[t * 16 * VS], ds, j, g);
    }
  }
  if (q0 + j < p.N) {
    float* op = p.dq + (row0 + q0 + j) * p.ld_dq + h * DH + 4 * g;
#pragma unroll
    for (int d = 0; d < DH / 16; ++d)
      *reinterpret_cast<float4*>(op + d * 16) = make_float4(dq[d][0] * p.scale, dq[d][1] * p.scale, dq[d][2] * p.scale, dq[d][3] * p.scale);
  }
}

template <int DH>
__global__ __launch_bounds__(256) void mha_bwd_dkv_mfma_kernel(MhaParams p) {
  constexpr int VS = DH + 4;
  __shared__ __attribute__((aligned(16))) float sQ[XT * DH], sG[XT * DH], sQv[XT * VS], sGv[XT * VS], sL[XT], sD[XT];
  __shared__ uint32_t sM[XT * BW], sP[XT * BW];         // rows = the staged queries, columns = this block's keys
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int tiles = (p.N + XT - 1) / XT;
  const int bh = blockIdx.x / tiles, kt = blockIdx.x - bh * tiles, b = bh / p.heads, h = bh - b * p.heads;
  const int key0 = kt * XT + wave * 16, key = min(key0 + j, p.N - 1);
  const bool key_live = key0 + j < p.N;
  const long long row0 = (long long)b * p.N, hrow = ((long long)b * p.heads + h) * p.N;
  float kreg[DH / 4], vreg[DH / 4];
  load_breg<DH>(kreg, p.k + (row0 + key) * p.ld_k + h * DH, g, 1.f);
  load_breg<DH>(vreg, p.v + (row0 + key) * p.ld_v + h * DH, g, 1.f);
  f32x4 dk[DH / 16], dv[DH / 16];
#pragma unroll
  for (int d = 0; d < DH / 16; ++d) dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}, dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  RowRegs<DH> rq, rg;
  ByteRegs bm, bp;
  float r_lse = 0.f, r_dsum = 0.f;
  auto fetch = [&](int i0) {
    load_rows<DH>(rq, p.q + row0 * p.ld_q + h * DH, p.ld_q, i0, p.N);
    load_rows<DH>(rg, p.dout + row0 * p.ld_do + h * DH, p.ld_do, i0, p.N);
    if (threadIdx.x < XT) {
      const bool in = i0 + (int)threadIdx.x < p.N;
      r_lse = in ? p.lse[hrow + i0 + threadIdx.x] : 0.f;
      r_dsum = in ? p.dsum[hrow + i0 + threadIdx.x] : 0.f;
    }
    if (p.mask) load_bytes(bm, p.mask + row0 * p.N, i0, kt * XT, p.N, 1);
    if (p.keep) load_bytes(bp, p.keep + hrow * p.N, i0, kt * XT, p.N, 0);
  };
  fetch(0);
  for (int i0 = 0; i0 < p.N; i0 += XT) {
    __syncthreads();
    store_score<DH>(sQ, rq, p.scale);
    store_score<DH>(sG, rg, 1.f);
    store_value<DH>(sQv, rq, p.scale);
    store_value<DH>(sGv, rg, 1.f);
    if (threadIdx.x < XT) sL[threadIdx.x] = r_lse, sD[threadIdx.x] = r_dsum;
    if (p.mask) store_bytes(sM, bm);
    if (p.keep) store_bytes(sP, bp);
    __syncthreads();
    if (i0 + XT < p.N) fetch(i0 + XT);
#pragma unroll
    for (int t = 0; t < XT / 16; ++t) {
      if (i0 + t * 16 >= p.N) break;
      const f32x4 s = score_tile<DH>(&sQ[t * 16 * DH], kreg, j, g);      // lane (key j, g): queries 4 g + r of the tile
      const f32x4 dp = score_tile<DH>(&sG[t * 16 * DH], vreg, j, g);
      const int ib = i0 + t * 16 + 4 * g;
      float w[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ib + r, lr = t * 16 + 4 * g + r;                      // query, its row in the staged tile
        const uint8_t mb = p.mask ? reinterpret_cast<const uint8_t*>(sM)[lr * (BW * 4) + wave * 16 + j] : (uint8_t)0;
        const uint8_t kb_ = p.keep ? reinterpret_cast<const uint8_t*>(sP)[lr * (BW * 4) + wave * 16 + j] : (uint8_t)1;
        const bool ok = i < p.N && key_live && !mb;
        const float pr = ok ? expf(s[r] - sL[lr]) : 0.f;
        w[r] = kb_ ? (p.keep ? pr * p.keep_scale : pr) : 0.f;
        const float d_ = kb_ ? (p.keep ? dp[r] * p.keep_scale : dp[r]) : 0.f;
        ds[r] = pr * (d_ - sD[lr]);
      }
      accum_tile<DH>(dv, &sGv[t * 16 * VS], w, j, g);
      accum_tile<DH>(dk, &sQv[t * 16 * VS], ds, j, g);                    // (sQv carries the scale)
    }
  }
  if (key_live) {
    float* kp = p.dk + (row0 + key0 + j) * p.ld_dk + h * DH + 4 * g;
    float* vp = p.dv + (row0 + key0 + j) * p.ld_dv + h * DH + 4 * g;
#pragma unroll
    for (int d = 0; d < DH / 16; ++d) {
      *reinterpret_cast<float4*>(kp + d * 16) = make_float4(dk[d][0], dk[d][1], dk[d][2], dk[d][3]);
      *reinterpret_cast<float4*>(vp + d * 16) = make_float4(dv[d][0], dv[d][1], dv[d][2], dv[d][3]);
    }
  }
}

// operands the MFMA kernels can take: float2 / float4 staging and float4 stores
static bool mfma_operands_ok(const MhaParams& p, int Dh, bool backward) {
  auto al = [](const void* q, long long ld) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0 && ld % 4 == 0; };
  if (Dh % 16 != 0) return false;
  if (!backward) return al(p.q, p.ld_q) && al(p.k, p.ld_k) && al(p.v, p.ld_v) && al(p.o, p.ld_o);
  return al(p.q, p.ld_q) && al(p.k, p.ld_k) && al(p.v, p.ld_v) && al(p.out, p.ld_o) && al(p.dout, p.ld_do) && al(p.dq, p.ld_dq) &&
         al(p.dk, p.ld_dk) && al(p.dv, p.ld_dv);
}

template <int DH>
int launch_mfma(const MhaParams& p, int B, bool backward, hipStream_t s) {
  const unsigned blocks = (unsigned)((long long)B * p.heads * ((p.N + XT - 1) / XT));
  ff3d_clear_error();
  if (!backward) {
    hipLaunchKernelGGL(mha_fwd_mfma_kernel<DH>, dim3(blocks), dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL(mha_bwd_dq_mfma_kernel<DH>, dim3(blocks), dim3(256), 0, s, p);
    hipLaunchKernelGGL(mha_bwd_dkv_mfma_kernel<DH>, dim3(blocks), dim3(256), 0, s, p);
  }
  return ff3d_launch_status();
}

template <int DH>
int launch_all(const MhaParams& p, int B, bool backward, hipStream_t s) {
  const unsigned blocks = (unsigned)((long long)B * p.heads * ((p.N + RB - 1) / RB));
  ff3d_clear_error();
  if (!backward) {
    hipLaunchKernelGGL(mha_fwd_kernel<DH>, dim3(blocks), dim3(TB), 0, s, p);
  } else {
    hipLaunchKernelGGL(mha_bwd_dq_kernel<DH>, dim3(blocks), dim3(TB), 0, s, p);
    hipLaunchKernelGGL(mha_bwd_dkv_kernel<DH>, dim3(blocks), dim3(TB), 0, s, p);
  }
  return ff3d_launch_status();
}

int dispatch(const MhaParams& p, int B, int Dh, bool backward, hipStream_t s) {
  static const bool scalar_only = [] {
    const char* e = getenv("FF3D_MHA_TRAIN_SCALAR");
    return e && e[0] == '1';
  }();
  if (!scalar_only && mfma_operands_ok(p, Dh, backward)) {
    switch (Dh) {
      case 16: return launch_mfma<16>(p, B, backward, s);
      case 32: return launch_mfma<32>(p, B, backward, s);
      case 64: return launch_mfma<64>(p, B, backward, s);
      default: break;
    }
  }
  switch (Dh) {
    case 4: return launch_all<4>(p, B, backward, s);
    case 8: return launch_all<8>(p, B, backward, s);
    case 16: return launch_all<16>(p, B, backward, s);
    case 32: return launch_all<32>(p, B, backward, s);
    case 64: return launch_all<64>(p, B, backward, s);
    default: return FF3D_ERR_UNSUPPORTED;
  }
}

}  // namespace

extern "C" int ff3d_mha_train_fwd(const float* q, const float* k, const float* v, const uint8_t* mask, const uint8_t* keep,
                                  float keep_scale, float* out, float* lse, int B, int N, int heads, int Dh, int64_t ld_q,
                                  int64_t ld_k, int64_t ld_v, int64_t ld_o, float scale, ff3d_stream_t stream) {
  FF3D_REQUIRE(q && k && v && out && lse, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && N > 0 && heads > 0 && (long long)B * heads * ((N + RB - 1) / RB) < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  const int64_t w = (int64_t)heads * Dh;
  FF3D_REQUIRE(ld_q >= w && ld_k >= w && ld_v >= w && ld_o >= w, FF3D_ERR_BAD_SHAPE);
  MhaParams p = {};
  p.q = q, p.k = k, p.v = v, p.mask = mask, p.keep = keep, p.o = out, p.lse_w = lse;
  p.ld_q = ld_q, p.ld_k = ld_k, p.ld_v = ld_v, p.ld_o = ld_o, p.N = N, p.heads = heads, p.scale = scale, p.keep_scale = keep_scale;
  return dispatch(p, B, Dh, false, static_cast<hipStream_t>(stream));
}

extern "C" int ff3d_mha_train_bwd(const float* q, const float* k, const float* v, const uint8_t* mask, const uint8_t* keep,
                                  float keep_scale, const float* out, const float* lse, const float* grad_out, float* grad_q,
                                  float* grad_k, float* grad_v, float* dsum_workspace, int B, int N, int heads, int Dh,
                                  int64_t ld_q, int64_t ld_k, int64_t ld_v, int64_t ld_o, int64_t ld_go, int64_t ld_gq,
                                  int64_t ld_gk, int64_t ld_gv, float scale, ff3d_stream_t stream) {
  FF3D_REQUIRE(q && k && v && out && lse && grad_out && grad_q && grad_k && grad_v && dsum_workspace, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && N > 0 && heads > 0 && (long long)B * heads * ((N + RB - 1) / RB) < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  const int64_t w = (int64_t)heads * Dh;
  FF3D_REQUIRE(ld_q >= w && ld_k >= w && ld_v >= w && ld_o >= w && ld_go >= w && ld_gq >= w && ld_gk >= w && ld_gv >= w,
               FF3D_ERR_BAD_SHAPE);
  MhaParams p = {};
  p.q = q, p.k = k, p.v = v, p.mask = mask, p.keep = keep, p.out = out, p.lse = lse, p.dout = grad_out;
  p.dq = grad_q, p.dk = grad_k, p.dv = grad_v, p.dsum_w = dsum_workspace, p.dsum = dsum_workspace;
  p.ld_q = ld_q, p.ld_k = ld_k, p.ld_v = ld_v, p.ld_o = ld_o, p.ld_do = ld_go, p.ld_dq = ld_gq, p.ld_dk = ld_gk, p.ld_dv = ld_gv;
  p.N = N, p.heads = heads, p.scale = scale, p.keep_scale = keep_scale;
  return dispatch(p, B, Dh, true, static_cast<hipStream_t>(stream));
}
