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
