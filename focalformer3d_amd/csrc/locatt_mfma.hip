// Local 9 x 9 window attention on the fp16 matrix cores with fp32-class accuracy (round 6) - LocalContextAttentionBlock
// (encoder_utils.py:109-163: similar -> softmax(. / sqrt(C)) -> weighting; the reference's own CUDA extension
// ops/locatt_ops/kernels.cuh:4-80 launches one block per pixel and re-reads every key 81 times).
//
// locatt.hip (rounds 2-5) keeps the 81 window scores of a pixel in registers and walks the channels with scalar FMAs out of an LDS
// halo tile: 1.6 ms per call at 8 frames x 256 channels x 180 x 180 - 12 % of the fp32 FMA rate, bound by LDS latency.  The work
// IS matrix-shaped once a ROW of 16 query pixels is taken together: for a window row dy
//     S_dy (16 x 32) = Q[y, x0 .. x0+15, :] (16 x C)  .  K[y+dy, x0-8 .. x0+23, :]^T (C x 32)          (band |x' - x| <= 4 used)
//     out  (16 x C) += P_dy (16 x 32, zero outside the band)  .  V[y+dy, x0-8 .. x0+23, :] (32 x C)
// i.e. a banded "flash attention" whose score tile is 9/32 dense.  Operands are the (hi, lo') fp16 pairs the neck's 1x1 GEMMs
// already produce (three MFMA passes per product, two fp32 accumulators, as in splitmm.hip); the probabilities are split the same
// way in registers.  864 MFMAs per 16 pixels: 14 M per call = 0.1 ms of MFMA issue.
//
// Layouts that make it work without a single LDS transpose:
//   * scores are computed TRANSPOSED (A operand = 16 key pixels, B operand = 16 query pixels), and the two key tiles of a window are
//     INTERLEAVED - row m' of key tile t is pixel xw0 + 8 (m' >> 2) + 4 t + (m' & 3) - so that the accumulator lane (x = lane & 15,
//     kq = lane >> 4) ends up holding S[x][xw0 + 8 kq + j], j = 4 t + r = 0 .. 7: exactly the eight consecutive k-slots the lane
//     feeds as the B operand of the second product.  Softmax: row maxima / sums over the lane's 72 values and two xor-shuffles.
//   * the second product contracts over key PIXELS, so V must be pixel-contiguous per channel: a pre-pass rewrites the value pair
//     NHWC -> NCHW with a zero border ((B, C, H + 12, Wp) fp16 planes; halo reads never leave the plane, no masking).  A V fragment
//     (16 channels x 32 pixels) is then one ds_read_b128 per lane with the same k-slot order.
//   * block = 4 waves = 8 rows x 16 pixels; a wave owns TWO rows (18 score tiles x 2 accumulators = 288 registers, one wave per
//     SIMD), so a key / value row fragment fetched from LDS serves both rows' windows; channels are walked in chunks of 32 (one MFMA
//     K-step): the (8 + 8) x 32-pixel halo of a chunk is 64 KB (both planes), double-buffered by the LDS DMA (128 KB).  Out-of-map
//     key pixels come from the zero row every pair plane carries (ff3d.h ZERO-ROW CONTRACT): their score is 0 and they take part
//     in the softmax, exactly like the reference (kernels.cuh:30-40); out-of-map values are the zero border.
//   * LDS slots are XOR-swizzled (f = {0, 2, 3, 1} of the 4-pixel / 4-channel group) on the DMA's source side and on the fragment
//     read, so that every ds_read_b128 service group touches 16 distinct 16-byte bank columns (see splitmm.hip).
// Output: the context as an NHWC (hi, lo') pair with the value's exponent (a convex combination of values never exceeds their
// maximum) - what the block's next 1x1 GEMM reads; no NCHW tensor, no transposing pass on either side of the attention.
#include <cstdlib>

#include "ff3d_common.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4 = __attribute__((ext_vector_type(4))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int LA_K = 9, LA_R = 4, LA_TY = 8, LA_TX = 16, LA_HY = LA_TY + 2 * LA_R, LA_HX = 32, LA_CC = 32;
constexpr int LA_PLANE = LA_HY * LA_HX * LA_CC;            // halves of one plane of one chunk buffer (32 KB)
constexpr int LA_BUF = 2 * LA_PLANE;                       // hi + lo' (64 KB)
constexpr int LA_PAD_TOP = LA_R, LA_PAD_LEFT = 8;
constexpr float LA_LO_SCALE = 2048.f, LA_LO_INV = 1.f / 2048.f;

struct LaParams {
  const _Float16 *q_hi, *q_lo, *k_hi, *k_lo;       // NHWC pair planes (B*H*W (+ zero row), C)
  const _Float16 *vp_hi, *vp_lo;                   // padded NCHW planes (B, C, Hp, Wp)
  _Float16 *o_hi, *o_lo;                           // NHWC pair planes (B*H*W, C)
  const int *q_exp, *k_exp;
  int B, C, H, W, Hp, Wp;
  float scale;
  unsigned k_zero;                                 // byte offset of the zero row of the key planes
};

__device__ __forceinline__ int la_swz(int group) { return (0x78 >> (2 * (group & 3))) & 3; }    // f = {0, 2, 3, 1}

__device__ __forceinline__ void la_glds16(const _Float16* base, unsigned byte_off, _Float16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(base) + byte_off,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---- pre-pass: NHWC pair plane (B, H, W, C) -> zero-bordered NCHW plane (B, C, Hp, Wp); grid (Hp, C / 32, B * 2 planes)
__global__ __launch_bounds__(256) void la_nchw_pad_kernel(const _Float16* __restrict__ src_hi, const _Float16* __restrict__ src_lo,
                                                          _Float16* __restrict__ dst_hi, _Float16* __restrict__ dst_lo, int C, int H,
                                                          int W, int Hp, int Wp) {
  extern __shared__ __attribute__((aligned(16))) _Float16 tile[];         // [32][Wp + 8]
  const int yp = blockIdx.x, c0 = blockIdx.y * 32, b = blockIdx.z >> 1, plane = blockIdx.z & 1;
  const _Float16* src = plane ? src_lo : src_hi;
  _Float16* dst = plane ? dst_lo : dst_hi;
  const int ld = Wp + 8, tid = threadIdx.x, y = yp - LA_PAD_TOP;
  for (int i = tid; i < 32 * ld / 8; i += 256) reinterpret_cast<uint4*>(tile)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (y >= 0 && y < H) {
    for (int i = tid; i < W * 4; i += 256) {
      const int px = i >> 2, q = i & 3;
      const half8 v = *reinterpret_cast<const half8*>(src + ((long long)(b * H + y) * W + px) * C + c0 + q * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) tile[(q * 8 + j) * ld + px + LA_PAD_LEFT] = v[j];
    }
  }
  __syncthreads();
  const int pieces = Wp / 8;
  for (int i = tid; i < 32 * pieces; i += 256) {
    const int ch = i / pieces, pc = i - ch * pieces;
    *reinterpret_cast<uint4*>(dst + ((long long)(b * C + c0 + ch) * Hp + yp) * Wp + pc * 8) =
        *reinterpret_cast<const uint4*>(tile + ch * ld + pc * 8);
  }
}

// ---- the attention: grid (x tiles * y tiles, B), 256 threads
// ABL (FF3D_BUILD_EXPERIMENTS, FF3D_LA_ABLATE; wrong results): 1 no MFMA, 2 DMA of the first chunk only, 4 no fragment reads, 8 no stores
template <int ABL = 0>
__global__ __launch_bounds__(256, 1) void locatt_mfma_kernel(LaParams p) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];          // [2 buffers][hi | lo'][LA_PLANE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int tiles_x = (p.W + LA_TX - 1) / LA_TX;
  const int x0 = (blockIdx.x % tiles_x) * LA_TX, y0 = (blockIdx.x / tiles_x) * LA_TY, b = blockIdx.y;
  const int n_chunks = p.C / LA_CC;

  // piece p = i * 256 + tid of a chunk buffer: plane = p >> 11, halo row = (p >> 7) & 15, column (pixel | channel) = (p >> 2) & 31,
  // 16-byte slot = p & 3; the slot holds source piece slot ^ f(column group)
  const int st_row = (tid >> 7), st_col = (tid >> 2) & 31, st_slot = tid & 3;      // row of piece i: 2 * (i & 7) + st_row
  auto stage_k = [&](int cc, int buf) {
    const int gx = x0 - 8 + st_col;
    const int qs = st_slot ^ la_swz(st_col >> 3);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 2 * (i & 7) + st_row, gy = y0 - LA_R + row;
      const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const unsigned off = ok ? (unsigned)((((long long)(b * p.H + gy) * p.W + gx) * p.C + cc * LA_CC + qs * 8) * 2) : p.k_zero;
      // window pixels 0-3 and 28-31 only ever meet masked score entries (|x' - x| > 4 for every query of the tile): their lanes
      // are switched off in the DMA (a quarter of the key bytes); whatever the LDS holds there lands in scores the band mask discards
      if (st_col >= 4 && st_col < 28) la_glds16(i < 8 ? p.k_hi : p.k_lo, off, lds + buf * LA_BUF + (i * 256 + wave * 64) * 8);
    }
  };
  auto stage_v = [&](int cc, int buf) {
    const int qs = st_slot ^ la_swz(st_col >> 2);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 2 * (i & 7) + st_row;
      const unsigned off = (unsigned)((((long long)(b * p.C + cc * LA_CC + st_col) * p.Hp + y0 + row) * p.Wp + x0 + qs * 8) * 2);
      la_glds16(i < 8 ? p.vp_hi : p.vp_lo, off, lds + buf * LA_BUF + (i * 256 + wave * 64) * 8);
    }
  };
  // query fragments of this wave's two rows for chunk cc: lane (x = x0 + fr, channels cc * 32 + kq * 8 ..)
  const int yw = y0 + 2 * wave;
  auto load_q = [&](int cc, half8 (&qh)[2], half8 (&ql)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bool ok = x0 + fr < p.W && yw + u < p.H;
      const long long o = ((long long)(b * p.H + min(yw + u, p.H - 1)) * p.W + min(x0 + fr, p.W - 1)) * p.C + cc * LA_CC + kq * 8;
      const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      qh[u] = ok ? *reinterpret_cast<const half8*>(p.q_hi + o) : z;
      ql[u] = ok ? *reinterpret_cast<const half8*>(p.q_lo + o) : z;
    }
  };

  // ------------------------------------------------------------------ phase 1: scores of both rows, all 9 window rows
  f32x4 sm[2][LA_K][2], sx[2][LA_K][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int d = 0; d < LA_K; ++d)
#pragma unroll
      for (int t = 0; t < 2; ++t) sm[u][d][t] = f32x4{0.f, 0.f, 0.f, 0.f}, sx[u][d][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // key fragment (A operand) of halo row r, tile t: row m' = fr is pixel xpix = 8 (fr >> 2) + 4 t + (fr & 3), channels kq * 8 ..
  const int kf_base = ((8 * (fr >> 2) + (fr & 3)) * 4 + (kq ^ la_swz(fr >> 2))) * 8;      // + t * 4 pixels (same swizzle group)
  half8 qh[2], ql[2], qh_n[2], ql_n[2];
  stage_k(0, 0);
  load_q(0, qh, ql);
  for (int cc = 0; cc < n_chunks; ++cc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                          // chunk cc has landed; every wave is done with buffer (cc + 1) & 1
    if (cc + 1 < n_chunks) {
      load_q(cc + 1, qh_n, ql_n);
      if (!(ABL & 2)) stage_k(cc + 1, (cc + 1) & 1);
    } else {
      stage_v(0, (cc + 1) & 1);                               // phase 2's first chunk under the last score chunk
    }
    const _Float16* kb = lds + (cc & 1) * LA_BUF + (2 * wave * LA_HX) * 32 + kf_base;
    // fragments of halo row r + 1 are fetched before the MFMAs of row r (one wave per SIMD: nothing else covers an LDS round trip)
    half8 kh[2][2], kl[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      kh[0][t] = *reinterpret_cast<const half8*>(kb + t * 4 * 32);
      kl[0][t] = *reinterpret_cast<const half8*>(kb + LA_PLANE + t * 4 * 32);
    }
#pragma unroll
    for (int r = 0; r < 10; ++r) {                            // halo rows 2 * wave + r serve row u's window row dy = r - u
      if (r + 1 < 10 && !(ABL & 4)) {
        const _Float16* rowp = kb + ((r + 1) * LA_HX) * 32;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          kh[(r + 1) & 1][t] = *reinterpret_cast<const half8*>(rowp + t * 4 * 32);
          kl[(r + 1) & 1][t] = *reinterpret_cast<const half8*>(rowp + LA_PLANE + t * 4 * 32);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int d = r - u;
        if (d < 0 || d >= LA_K || (ABL & 1)) continue;
#pragma unroll
        for (int t = 0; t < 2; ++t) sm[u][d][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[r & 1][t], qh[u], sm[u][d][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) sx[u][d][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[r & 1][t], qh[u], sx[u][d][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) sx[u][d][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[r & 1][t], ql[u], sx[u][d][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (cc + 1 < n_chunks) {
#pragma unroll
      for (int u = 0; u < 2; ++u) qh[u] = qh_n[u], ql[u] = ql_n[u];
    }
  }

  // ------------------------------------------------------------------ softmax over the 9 x 9 window, per query pixel
  // lane (x = x0 + fr, kq): sm[u][d][t][r] = S[x][x' = x0 - 8 + 8 kq + 4 t + r]; in the band iff dx = 8 kq + 4 t + r - 4 - fr in [0, 8]
  const float s_scale = ff3d_pow2(ff3d_ld_exp(p.q_exp) + ff3d_ld_exp(p.k_exp)) * p.scale;
  half8 ph[2][LA_K], pl[2][LA_K];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float mx = -INFINITY;
#pragma unroll
    for (int d = 0; d < LA_K; ++d)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int dx = 8 * kq + 4 * t + r - 4 - fr;
          const float s = (sm[u][d][t][r] + sx[u][d][t][r] * LA_LO_INV) * s_scale;
          sm[u][d][t][r] = (dx >= 0 && dx < LA_K) ? s : -INFINITY;
          mx = fmaxf(mx, sm[u][d][t][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int d = 0; d < LA_K; ++d)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = expf(sm[u][d][t][r] - mx);           // exp(-inf) = 0 outside the band
          sm[u][d][t][r] = e;
          sum += e;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
#pragma unroll
    for (int d = 0; d < LA_K; ++d)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = sm[u][d][t][r] * inv;
          const _Float16 h = (_Float16)pv;
          ph[u][d][4 * t + r] = h;
          pl[u][d][4 * t + r] = (_Float16)((pv - (float)h) * LA_LO_SCALE);
        }
  }

  // ------------------------------------------------------------------ phase 2: context, 32 output channels per chunk
  // value fragment (A operand) of halo row r, channel tile ct: row m = fr is channel 8 (fr >> 2) + 4 ct + (fr & 3) of the chunk (the two
  // tiles interleaved, as the key tiles are: the accumulator lane then holds the 8 CONSECUTIVE channels 8 kq .. 8 kq + 7 of its pixel
  // over ct, r - one 16-byte store per plane); k-slots = pixels 8 kq .. 8 kq + 7
  int vf_base[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int ch = 8 * (fr >> 2) + 4 * ct + (fr & 3);
    vf_base[ct] = (ch * 4 + (kq ^ la_swz(ch >> 2))) * 8;
  }
  for (int cc = 0; cc < n_chunks; ++cc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int buf = (n_chunks + cc) & 1;                     // value chunk cc sits in buffer (n_chunks + cc) & 1
    if (cc + 1 < n_chunks && !(ABL & 2)) stage_v(cc + 1, buf ^ 1);
    const _Float16* vb = lds + buf * LA_BUF + (2 * wave * 32) * 32;
    f32x4 om[2][2], ox[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) om[u][ct] = f32x4{0.f, 0.f, 0.f, 0.f}, ox[u][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    half8 vh[2][2], vl[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      vh[0][ct] = *reinterpret_cast<const half8*>(vb + vf_base[ct]);
      vl[0][ct] = *reinterpret_cast<const half8*>(vb + LA_PLANE + vf_base[ct]);
    }
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      if (r + 1 < 10 && !(ABL & 4)) {
        const _Float16* rowp = vb + ((r + 1) * 32) * 32;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          vh[(r + 1) & 1][ct] = *reinterpret_cast<const half8*>(rowp + vf_base[ct]);
          vl[(r + 1) & 1][ct] = *reinterpret_cast<const half8*>(rowp + LA_PLANE + vf_base[ct]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int d = r - u;
        if (d < 0 || d >= LA_K || (ABL & 1)) continue;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) om[u][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[r & 1][ct], ph[u][d], om[u][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) ox[u][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[r & 1][ct], ph[u][d], ox[u][ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) ox[u][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[r & 1][ct], pl[u][d], ox[u][ct], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // lane: pixel x0 + fr of row yw + u, channels cc * 32 + 8 kq + (4 ct + r) -> one 16-byte store per plane
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (x0 + fr >= p.W || yw + u >= p.H || (ABL & 8)) continue;
      const long long o = ((long long)(b * p.H + yw + u) * p.W + x0 + fr) * p.C + cc * LA_CC + kq * 8;
      half8 h, l;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = om[u][ct][r] + ox[u][ct][r] * LA_LO_INV;
          h[4 * ct + r] = (_Float16)v;
          l[4 * ct + r] = (_Float16)((v - (float)h[4 * ct + r]) * LA_LO_SCALE);
        }
      *reinterpret_cast<half8*>(p.o_hi + o) = h;
      *reinterpret_cast<half8*>(p.o_lo + o) = l;
    }
  }
}

}  // namespace

extern "C" int64_t ff3d_local_attention_pair_workspace_halfs(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  const long long Hp = (long long)((H + LA_TY - 1) / LA_TY) * LA_TY + 2 * LA_R;
  const long long Wp = (long long)((W + LA_TX - 1) / LA_TX) * LA_TX + 16;
  return 2 * (long long)B * C * Hp * Wp;                   // hi plane, then lo' plane
}

extern "C" int ff3d_local_attention_pair(const void* q_hi, const void* q_lo, const int32_t* q_exp, const void* k_hi, const void* k_lo,
                                         const int32_t* k_exp, const void* v_hi, const void* v_lo, void* workspace, void* out_hi,
                                         void* out_lo, int B, int C, int H, int W, int k, float scale, ff3d_stream_t stream) {
  FF3D_REQUIRE(q_hi && q_lo && k_hi && k_lo && v_hi && v_lo && workspace && out_hi && out_lo, FF3D_ERR_NULL);
  FF3D_REQUIRE(k == LA_K, FF3D_ERR_UNSUPPORTED);
  FF3D_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % LA_CC == 0 && B <= 65535 / 2, FF3D_ERR_BAD_SHAPE);
  const int Hp = (H + LA_TY - 1) / LA_TY * LA_TY + 2 * LA_R, Wp = (W + LA_TX - 1) / LA_TX * LA_TX + 16;
  FF3D_REQUIRE(((long long)B * H * W + 1) * C * 2 < (1ll << 32) && (long long)B * C * Hp * Wp * 2 < (1ll << 32), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(q_hi) && ff3d_aligned16(q_lo) && ff3d_aligned16(k_hi) && ff3d_aligned16(k_lo) && ff3d_aligned16(v_hi) &&
                   ff3d_aligned16(v_lo) && ff3d_aligned16(workspace) && ff3d_aligned16(out_hi) && ff3d_aligned16(out_lo),
               FF3D_ERR_ALIGNMENT);
  hipStream_t s = static_cast<hipStream_t>(stream);
  _Float16* vp_hi = static_cast<_Float16*>(workspace);
  _Float16* vp_lo = vp_hi + (long long)B * C * Hp * Wp;
  ff3d_clear_error();
  const size_t tile_bytes = (size_t)32 * (Wp + 8) * sizeof(_Float16);
  FF3D_REQUIRE(tile_bytes <= 64 * 1024, FF3D_ERR_BAD_SHAPE);
  hipLaunchKernelGGL(la_nchw_pad_kernel, dim3(Hp, C / 32, B * 2), dim3(256), tile_bytes, s, static_cast<const _Float16*>(v_hi),
                     static_cast<const _Float16*>(v_lo), vp_hi, vp_lo, C, H, W, Hp, Wp);
  LaParams p;
  p.q_hi = static_cast<const _Float16*>(q_hi), p.q_lo = static_cast<const _Float16*>(q_lo);
  p.k_hi = static_cast<const _Float16*>(k_hi), p.k_lo = static_cast<const _Float16*>(k_lo);
  p.vp_hi = vp_hi, p.vp_lo = vp_lo;
  p.o_hi = static_cast<_Float16*>(out_hi), p.o_lo = static_cast<_Float16*>(out_lo);
  p.q_exp = q_exp, p.k_exp = k_exp;
  p.B = B, p.C = C, p.H = H, p.W = W, p.Hp = Hp, p.Wp = Wp;
  p.scale = scale;
  p.k_zero = (unsigned)((long long)B * H * W * C * 2);
  constexpr size_t lds_bytes = 2 * LA_BUF * sizeof(_Float16);        // 128 KiB
  static bool configured[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!configured[dev & 63]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&locatt_mfma_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess)
      return FF3D_ERR_LAUNCH;
    configured[dev & 63] = true;
  }
  const int tiles = ((W + LA_TX - 1) / LA_TX) * ((H + LA_TY - 1) / LA_TY);
#ifdef FF3D_BUILD_EXPERIMENTS
  static const int abl = [] {
    const char* e = getenv("FF3D_LA_ABLATE");
    return e ? atoi(e) : 0;
  }();
#define FF3D_LA(n)                                                                                                              \
  case n:                                                                                                                       \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&locatt_mfma_kernel<n>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds_bytes);                                                                                  \
    hipLaunchKernelGGL(locatt_mfma_kernel<n>, dim3(tiles, B), dim3(256), lds_bytes, s, p);                                      \
    return ff3d_launch_status();
  switch (abl) { FF3D_LA(1) FF3D_LA(2) FF3D_LA(4) FF3D_LA(8) FF3D_LA(3) FF3D_LA(6) FF3D_LA(7) FF3D_LA(15) default: break; }
#undef FF3D_LA
#endif
  hipLaunchKernelGGL(locatt_mfma_kernel<0>, dim3(tiles, B), dim3(256), lds_bytes, s, p);
  return ff3d_launch_status();
}
