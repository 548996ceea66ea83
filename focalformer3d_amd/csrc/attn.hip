// Query self-attention core (softmax(Q K^T / sqrt(Dh)) V per (frame, head)) for gfx950, exact fp32 on the
// matrix cores: v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate, bitwise an fmaf chain).
//
// Shape regime: N = 200..1000 queries, Dh = 16/32, B*heads = 8..256 independent problems - far too small for
// a library GEMM pair plus a softmax kernel, so everything is fused flash-style with the score matrix never
// leaving registers:
//   * a 256-thread block owns 64 queries of one (frame, head); each of its 4 waves a 16-query tile;
//   * K/V arrive in 64-key tiles through LDS (coalesced 8/16-byte global loads), laid out so the MFMA operand
//     reads are bank-conflict free: K as [dim/2][key][2] (lanes of a half-wave hit 32 distinct banks), V
//     row-major with a row stride of Dh+4 floats;
//   * scores are computed TRANSPOSED (S^T = K Q^T), so a lane holds 4 keys of one query: the softmax
//     reduction over keys is 3 register ops + two cross-lane xor-shuffles (16, 32), and exp(S^T) is directly
//     the B operand of O^T += V^T P^T with the 4 MFMAs of a tile taking keys {4g+r} - no LDS round trip,
//     no transposition of P;
//   * online softmax (running max / sum per query) in registers; O^T accumulators are rescaled in place.
#include "ff3d_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct AttnParams {
  const float *q, *k, *v;
  float* out;
  long long ld_q, ld_k, ld_v, ld_o;
  int N, heads, Dh;
  float scale;
};

constexpr int KT = 64;  // keys staged per iteration

template <int DH>
__global__ __launch_bounds__(256) void self_attn_mfma_kernel(AttnParams p) {
  constexpr int VS = DH + 4;                     // V row stride (floats): 4*VS % 32 == 16 -> conflict-free
  __shared__ __attribute__((aligned(16))) float sK[KT * DH];   // [tile 0..3][dim/2][16 keys][2]
  __shared__ __attribute__((aligned(16))) float sV[KT * VS];   // [key][VS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int qtiles = (p.N + 63) / 64;
  const int bh = blockIdx.x / qtiles, qt = blockIdx.x - bh * qtiles;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int q0 = qt * 64 + wave * 16;
  const long long row0 = (long long)b * p.N;

  // Q^T B-operand: B[k=g][j] = Q[q0+j][4c+g], pre-scaled
  float qreg[DH / 4];
  {
    const int qi = min(q0 + j, p.N - 1);
    const float* qp = p.q + (row0 + qi) * p.ld_q + h * DH;
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) qreg[c] = qp[4 * c + g] * p.scale;
  }
  f32x4 o[DH / 16];
#pragma unroll
  for (int d = 0; d < DH / 16; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  for (int k0 = 0; k0 < p.N; k0 += KT) {
    __syncthreads();   // previous tile fully consumed
    // ---- stage K: float2 (dims 2u, 2u+1) of key kk -> sK[(kk>>4)*16*DH + u*32 + (kk&15)*2]
    for (int e = tid; e < KT * DH / 2; e += 256) {
      const int kk = e / (DH / 2), u = e - kk * (DH / 2);
      float2 val = make_float2(0.f, 0.f);
      if (k0 + kk < p.N) val = *reinterpret_cast<const float2*>(p.k + (row0 + k0 + kk) * p.ld_k + h * DH + 2 * u);
      *reinterpret_cast<float2*>(&sK[(kk >> 4) * 16 * DH + u * 32 + (kk & 15) * 2]) = val;
    }
    // ---- stage V: float4 chunks, row-major with padded stride
    for (int e = tid; e < KT * DH / 4; e += 256) {
      const int kk = e / (DH / 4), u = e - kk * (DH / 4);
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + kk < p.N) val = *reinterpret_cast<const float4*>(p.v + (row0 + k0 + kk) * p.ld_v + h * DH + 4 * u);
      *reinterpret_cast<float4*>(&sV[kk * VS + 4 * u]) = val;
    }
    __syncthreads();

#pragma unroll
    for (int t = 0; t < KT / 16; ++t) {
      if (k0 + t * 16 >= p.N) break;             // wave-uniform
      // S^T tile (16 keys x 16 queries): A[i=key][k] = K[key][4c+k], B[k][j] = Q^T
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* kt = &sK[t * 16 * DH];
#pragma unroll
      for (int c = 0; c < DH / 4; ++c) {
        const int dim = 4 * c + g;
        const float a = kt[(dim >> 1) * 32 + j * 2 + (dim & 1)];   // lane (i=j, k=g): key j of the tile
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qreg[c], s, 0, 0, 0);
      }
      // lane (query j, g): s[r] = score of key k0 + t*16 + 4g + r
      const int kbase = k0 + t * 16 + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (kbase + r >= p.N) s[r] = -INFINITY;
      float m_loc = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
      m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 16));
      m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 32));
      const float m_new = fmaxf(m_run, m_loc);     // finite: every tile has >= 1 valid key
      const float alpha = expf(m_run - m_new);     // 0 on the first tile (m_run = -inf)
      float pr[4];
      float l_loc = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[r] = expf(s[r] - m_new);
        l_loc += pr[r];
      }
      l_loc += __shfl_xor(l_loc, 16);
      l_loc += __shfl_xor(l_loc, 32);
      l_run = l_run * alpha + l_loc;
      m_run = m_new;
      // O^T (16 dims x 16 queries) = alpha * O^T + V^T P^T ; MFMA r contracts over keys {4g + r}
      const float* vt = &sV[(t * 16) * VS];
#pragma unroll
      for (int d = 0; d < DH / 16; ++d) {
        f32x4 acc = o[d];
        acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = vt[(4 * g + r) * VS + d * 16 + j];     // A[i=dim j][k=g] = V[key 4g+r][dim]
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pr[r], acc, 0, 0, 0);
        }
        o[d] = acc;
      }
    }
  }
  // lane (query j, g) holds O^T rows 4g..4g+3 (dims) of each 16-dim block
  if (q0 + j < p.N) {
    const float inv = 1.f / l_run;
    float* op = p.out + (row0 + q0 + j) * p.ld_o + h * DH + 4 * g;
#pragma unroll
    for (int d = 0; d < DH / 16; ++d)
      *reinterpret_cast<float4*>(op + d * 16) = make_float4(o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv);
  }
}

// Tiny heads (Dh not a multiple of 16; test-size models): one thread per (frame, head, query), two passes.
__global__ __launch_bounds__(128) void self_attn_small_kernel(AttnParams p, int total) {
  const int idx = blockIdx.x * 128 + threadIdx.x;
  if (idx >= total) return;
  const int qi = idx % p.N, bh = idx / p.N;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const long long row0 = (long long)b * p.N;
  const float* qp = p.q + (row0 + qi) * p.ld_q + h * p.Dh;
  float m = -INFINITY;
  for (int kk = 0; kk < p.N; ++kk) {
    const float* kp = p.k + (row0 + kk) * p.ld_k + h * p.Dh;
    float s = 0.f;
    for (int d = 0; d < p.Dh; ++d) s = fmaf(qp[d] * p.scale, kp[d], s);
    m = fmaxf(m, s);
  }
  float l = 0.f;
  float acc[64];
  for (int d = 0; d < p.Dh; ++d) acc[d] = 0.f;
  for (int kk = 0; kk < p.N; ++kk) {
    const float* kp = p.k + (row0 + kk) * p.ld_k + h * p.Dh;
    const float* vp = p.v + (row0 + kk) * p.ld_v + h * p.Dh;
    float s = 0.f;
    for (int d = 0; d < p.Dh; ++d) s = fmaf(qp[d] * p.scale, kp[d], s);
    const float w = expf(s - m);
    l += w;
    for (int d = 0; d < p.Dh; ++d) acc[d] = fmaf(w, vp[d], acc[d]);
  }
  float* op = p.out + (row0 + qi) * p.ld_o + h * p.Dh;
  for (int d = 0; d < p.Dh; ++d) op[d] = acc[d] / l;
}

}  // namespace

extern "C" int ff3d_self_attention(const float* q, const float* k, const float* v, float* out, int B, int N, int heads,
                                   int Dh, int64_t ld_q, int64_t ld_k, int64_t ld_v, int64_t ld_o, float scale,
                                   ff3d_stream_t stream) {
  FF3D_REQUIRE(q && k && v && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && N > 0 && heads > 0 && Dh > 0 && Dh <= 64, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ld_q >= (int64_t)heads * Dh && ld_k >= (int64_t)heads * Dh && ld_v >= (int64_t)heads * Dh &&
                   ld_o >= (int64_t)heads * Dh,
               FF3D_ERR_BAD_SHAPE);
  AttnParams p{q, k, v, out, ld_q, ld_k, ld_v, ld_o, N, heads, Dh, scale};
  hipStream_t s = static_cast<hipStream_t>(stream);
  ff3d_clear_error();
  if (Dh % 16 == 0) {
    FF3D_REQUIRE(ff3d_aligned16(k) && ff3d_aligned16(v) && ff3d_aligned16(out) && ld_k % 4 == 0 && ld_v % 4 == 0 &&
                     ld_o % 4 == 0,
                 FF3D_ERR_ALIGNMENT);
    const long long blocks = (long long)B * heads * ((N + 63) / 64);
    FF3D_REQUIRE(blocks < (1ll << 31), FF3D_ERR_BAD_SHAPE);
    switch (Dh) {
      case 16: hipLaunchKernelGGL(self_attn_mfma_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, p); break;
      case 32: hipLaunchKernelGGL(self_attn_mfma_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, s, p); break;
      case 48: hipLaunchKernelGGL(self_attn_mfma_kernel<48>, dim3((unsigned)blocks), dim3(256), 0, s, p); break;
      case 64: hipLaunchKernelGGL(self_attn_mfma_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, s, p); break;
      default: return FF3D_ERR_BAD_SHAPE;
    }
  } else {
    const long long total = (long long)B * heads * N;
    FF3D_REQUIRE(total < (1ll << 31), FF3D_ERR_BAD_SHAPE);
    hipLaunchKernelGGL(self_attn_small_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, s, p, (int)total);
  }
  return ff3d_launch_status();
}
