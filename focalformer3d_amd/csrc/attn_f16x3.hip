// Query self-attention core (softmax(Q K^T / sqrt(Dh)) V per (frame, head)) on the fp16 matrix cores with fp32-class
// accuracy: the flash structure of attn.hip with every fp32 operand carried as a (hi, lo') fp16 pair and three
// v_mfma_f32_16x16x32_f16 passes per product (splitmm.hip arithmetic) - 24 MFMAs of 16 cycles per 64-key tile and wave
// instead of 64 fp32 MFMAs of 32 cycles.
//   * block = 64 queries of one (frame, head), wave = 16 queries; K / V arrive in 64-key tiles, converted to pairs while
//     they are staged into LDS: K as [key][32 dims] rows (64 B, XOR chunk swizzle of splitmm.hip; Dh = 16 zero-padded),
//     V transposed as [dim][key] with the keys of every 32-key block stored in the order pos = 8*((k>>2)&3) + 4*(k>>4) + (k&3);
//   * scores are computed transposed (S^T = K Q^T: lane (query j, g) holds keys 16t + 4g + r), two 16-key tiles per
//     32-key block; with the permutation above the eight probabilities a lane holds for a block ARE its B fragment of
//     O^T += V^T P^T (MFMA k index 8g + 4t + r) - no LDS round trip, no transposition of P;
//   * online softmax per 32-key block in registers (max / sum over keys = in-lane ops + xor-shuffles 16, 32).
#include <cstdlib>

#include "ff3d_common.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct AttnF16Params {
  const float *q, *k, *v;
  float* out;
  long long ld_q, ld_k, ld_v, ld_o;
  int N, heads;
  float scale;
};

__device__ __forceinline__ int at_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

__device__ __forceinline__ void at_split(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)((x - (float)hi) * 2048.f);
}

// NW waves per block = 16 * NW queries of one (frame, head): every block converts and stages the whole K / V of its head, so
// larger query tiles amortise that VALU + LDS-write work (the kernel is bound by it, not by its MFMAs).
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW) void self_attn_f16x3_kernel(AttnF16Params p) {
  constexpr int T = 64 * NW, QT = 16 * NW;
  constexpr int VROW = 72;                               // halves per V^T row: 64 keys + 8 pad (144-byte stride)
  __shared__ __attribute__((aligned(16))) _Float16 sK[2][64 * 32];       // [plane][key][32 dims]
  __shared__ __attribute__((aligned(16))) _Float16 sVt[2][DH * VROW];    // [plane][dim][permuted key]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int qtiles = (p.N + QT - 1) / QT;
  const int bh = blockIdx.x / qtiles, qt = blockIdx.x - bh * qtiles;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int q0 = qt * QT + wave * 16;
  const long long row0 = (long long)b * p.N;

  // Q^T fragment (B operand of S^T = K Q^T): lane (query fr, kq) holds dims 8*kq .. 8*kq + 7, pre-scaled
  half8 qh, ql;
  {
    const int qi = min(q0 + fr, p.N - 1);
    const float* qp = p.q + (row0 + qi) * p.ld_q + h * DH;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int d = kq * 8 + i;
      _Float16 hh, ll;
      at_split(d < DH ? qp[d] * p.scale : 0.f, hh, ll);
      qh[i] = hh, ql[i] = ll;
    }
  }
  f32x4 om[DH / 16], ox[DH / 16];
#pragma unroll
  for (int d = 0; d < DH / 16; ++d) om[d] = f32x4{0.f, 0.f, 0.f, 0.f}, ox[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  // K / V of a 64-key tile: 64 * 8 float4 each = NPRE per thread.  Round 5: the NEXT tile's values are fetched into registers
  // while the current tile is computed (rounds 1-4 loaded, converted and staged a tile between two barriers: ten exposed global
  // round trips per block at 600 keys)
  constexpr int NPRE = (64 * 8 + T - 1) / T, NPREV = (64 * DH / 4 + T - 1) / T;
  float4 kpre[NPRE], vpre[NPREV];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int e = i * T + tid, kk = e >> 3, u = e & 7;
      kpre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < 64 * 8 && 4 * u < DH && k0 + kk < p.N)
        kpre[i] = *reinterpret_cast<const float4*>(p.k + (row0 + k0 + kk) * p.ld_k + h * DH + 4 * u);
    }
#pragma unroll
    for (int i = 0; i < NPREV; ++i) {
      const int e = i * T + tid, kk = e / (DH / 4), u = e - kk * (DH / 4);
      vpre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < 64 * DH / 4 && k0 + kk < p.N)
        vpre[i] = *reinterpret_cast<const float4*>(p.v + (row0 + k0 + kk) * p.ld_v + h * DH + 4 * u);
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < p.N; k0 += 64) {
    __syncthreads();                                     // previous tile fully consumed
    // ---- stage K: 4 dims of key kk per step -> 8 bytes of the hi and lo rows (dims >= DH are zero)
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int e = i * T + tid;
      if (e >= 64 * 8) break;
      const int kk = e >> 3, u = e & 7;
      const float4 val = kpre[i];
      _Float16 hh[4], ll[4];
      at_split(val.x, hh[0], ll[0]);
      at_split(val.y, hh[1], ll[1]);
      at_split(val.z, hh[2], ll[2]);
      at_split(val.w, hh[3], ll[3]);
      const int o = kk * 32 + (((u >> 1) ^ at_swz(kk)) * 8) + (u & 1) * 4;
      *reinterpret_cast<uint2*>(&sK[0][o]) = *reinterpret_cast<uint2*>(hh);
      *reinterpret_cast<uint2*>(&sK[1][o]) = *reinterpret_cast<uint2*>(ll);
    }
    // ---- stage V^T with the key permutation of the header
#pragma unroll
    for (int i = 0; i < NPREV; ++i) {
      const int e = i * T + tid;
      if (e >= 64 * DH / 4) break;
      const int kk = e / (DH / 4), u = e - kk * (DH / 4);
      const float4 val = vpre[i];
      const int k5 = kk & 31, pos = (kk & 32) + ((k5 >> 2) & 3) * 8 + ((k5 >> 4) & 1) * 4 + (k5 & 3);
      const float f[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        _Float16 hh, ll;
        at_split(f[c], hh, ll);
        sVt[0][(4 * u + c) * VROW + pos] = hh;
        sVt[1][(4 * u + c) * VROW + pos] = ll;
      }
    }
    if (k0 + 64 < p.N) fetch(k0 + 64);                   // in flight under this tile's MFMAs and softmax
    __syncthreads();

#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      if (k0 + blk * 32 >= p.N) break;                   // wave-uniform
      float s[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = blk * 32 + t * 16 + fr;          // A row = key of the tile
        const int o = row * 32 + ((kq ^ at_swz(row)) * 8);
        const half8 ah = *reinterpret_cast<const half8*>(&sK[0][o]);
        const half8 al = *reinterpret_cast<const half8*>(&sK[1][o]);
        f32x4 sm = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, qh, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        f32x4 sx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ql, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        sx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, qh, sx, 0, 0, 0);
        const int kbase = k0 + blk * 32 + t * 16 + 4 * kq;   // lane (query fr, kq): s[t][r] = score of key kbase + r
#pragma unroll
        for (int r = 0; r < 4; ++r) s[t][r] = (kbase + r < p.N) ? sm[r] + sx[r] * (1.f / 2048.f) : -INFINITY;
      }
      float m_loc = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])),
                          fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
      m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 16));
      m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 32));
      const float m_new = fmaxf(m_run, m_loc);           // finite: the block's first key is valid
      const float alpha = expf(m_run - m_new);           // 0 on the first block (m_run = -inf)
      half8 ph, pl;                                      // P^T fragment: MFMA k index 8*kq + 4*t + r
      float l_loc = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = expf(s[t][r] - m_new);
          l_loc += pr;
          _Float16 hh, ll;
          at_split(pr, hh, ll);
          ph[4 * t + r] = hh, pl[4 * t + r] = ll;
        }
      l_loc += __shfl_xor(l_loc, 16);
      l_loc += __shfl_xor(l_loc, 32);
      l_run = l_run * alpha + l_loc;
      m_run = m_new;
#pragma unroll
      for (int d = 0; d < DH / 16; ++d) {
        const int o = (d * 16 + fr) * VROW + blk * 32 + kq * 8;     // A row = dim, 8 permuted keys
        const half8 vh = *reinterpret_cast<const half8*>(&sVt[0][o]);
        const half8 vl = *reinterpret_cast<const half8*>(&sVt[1][o]);
        f32x4 am = om[d], ax = ox[d];
#pragma unroll
        for (int r = 0; r < 4; ++r) am[r] *= alpha, ax[r] *= alpha;
        am = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, am, 0, 0, 0);
        ax = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, ax, 0, 0, 0);
        ax = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, ax, 0, 0, 0);
        om[d] = am, ox[d] = ax;
      }
    }
  }
  // lane (query fr, kq) holds O^T rows 4*kq .. 4*kq + 3 (dims) of each 16-dim block
  if (q0 + fr < p.N) {
    const float inv = 1.f / l_run;
    float* op = p.out + (row0 + q0 + fr) * p.ld_o + h * DH + 4 * kq;
#pragma unroll
    for (int d = 0; d < DH / 16; ++d)
      *reinterpret_cast<float4*>(op + d * 16) =
          make_float4((om[d][0] + ox[d][0] * (1.f / 2048.f)) * inv, (om[d][1] + ox[d][1] * (1.f / 2048.f)) * inv,
                      (om[d][2] + ox[d][2] * (1.f / 2048.f)) * inv, (om[d][3] + ox[d][3] * (1.f / 2048.f)) * inv);
  }
}

}  // namespace

extern "C" int ff3d_self_attention_f16x3(const float* q, const float* k, const float* v, float* out, int B, int N,
                                         int heads, int Dh, int64_t ld_q, int64_t ld_k, int64_t ld_v, int64_t ld_o,
                                         float scale, ff3d_stream_t stream) {
  FF3D_REQUIRE(q && k && v && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && N > 0 && heads > 0 && (Dh == 16 || Dh == 32), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ld_q >= (int64_t)heads * Dh && ld_k >= (int64_t)heads * Dh && ld_v >= (int64_t)heads * Dh &&
                   ld_o >= (int64_t)heads * Dh,
               FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(k) && ff3d_aligned16(v) && ff3d_aligned16(out) && ld_k % 4 == 0 && ld_v % 4 == 0 && ld_o % 4 == 0,
               FF3D_ERR_ALIGNMENT);
  static const int nw_env = [] {                  // tuning hook: FF3D_ATTN_NW = 4 | 8 waves per block
    const char* e = getenv("FF3D_ATTN_NW");
    return e ? atoi(e) : 0;
  }();
  // 8 waves (128 queries) per block: 0.117 vs 0.150 ms at 32 frames x 600 queries, 0.032 vs 0.042 at 4 frames, equal at 1
  // (profiles/r02_g_attention_block_size.txt); sequences of at most 64 queries keep the 4-wave block
  const int nw = nw_env == 4 || nw_env == 8 ? nw_env : (N > 64 ? 8 : 4);
  const long long blocks = (long long)B * heads * ((N + 16 * nw - 1) / (16 * nw));
  FF3D_REQUIRE(blocks < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  AttnF16Params p{q, k, v, out, ld_q, ld_k, ld_v, ld_o, N, heads, scale};
  hipStream_t s = static_cast<hipStream_t>(stream);
  ff3d_clear_error();
  if (Dh == 32 && nw == 8)
    hipLaunchKernelGGL((self_attn_f16x3_kernel<32, 8>), dim3((unsigned)blocks), dim3(512), 0, s, p);
  else if (Dh == 32)
    hipLaunchKernelGGL((self_attn_f16x3_kernel<32, 4>), dim3((unsigned)blocks), dim3(256), 0, s, p);
  else if (nw == 8)
    hipLaunchKernelGGL((self_attn_f16x3_kernel<16, 8>), dim3((unsigned)blocks), dim3(512), 0, s, p);
  else
    hipLaunchKernelGGL((self_attn_f16x3_kernel<16, 4>), dim3((unsigned)blocks), dim3(256), 0, s, p);
  return ff3d_launch_status();
}
