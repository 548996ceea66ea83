// Final layer of the heatmap head for gfx950: out = conv3x3(relu(x + in_bias), w) + bias with a SMALL number of
// output channels (K <= 16: 10 nuScenes / 3 Waymo classes), exact fp32 on the matrix cores.
//
// The vendor path spends three launches here (channel-shift add, ReLU, a Winograd conv tuned for wide outputs);
// this is one implicit-GEMM kernel: M = 16 consecutive pixels of a row, N = 16 (classes, zero padded), K = 4 input
// channels of one filter tap per v_mfma_f32_16x16x4_f32.  A 256-thread block owns an 8 x 32 pixel tile; per chunk of
// 16 input channels the halo tile is staged ONCE through LDS with the folded-BatchNorm shift + ReLU applied on the
// way in (so that pass over the (B,C,H,W) activation disappears), the chunk's weights are staged as
// [tap][channel][class]; each wave keeps 4 accumulators (4 pixel groups) so every weight fragment feeds 4 MFMAs.
// The LDS tile is linear in staging order (immediate-offset stores; operand reads are at worst 2-way conflicted on
// 4 of 32 lanes) and the chunk-invariant staging geometry lives in 24 registers, so a chunk's staging is ~5
// instructions per element; operand fetch for tap t+1 is issued before the MFMAs of tap t.
#include "ff3d_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int CT_Y = 8, CT_X = 32, CCH = 16;

struct ConvHeadParams {
  const float *x, *in_bias, *w, *bias;
  float* out;
  int C, H, W, K, relu;
};

__global__ __launch_bounds__(256) void relu_conv3x3_small_kernel(ConvHeadParams p) {
  constexpr int HX = CT_X + 2, HALO = (CT_Y + 2) * HX;          // 34, 340 elements per channel
  constexpr int NEL = CCH * HALO;                               // 5440 floats per chunk
  constexpr int NIN = (NEL + 255) / 256;                        // 22 staging slots per thread per chunk
  __shared__ float s_in[NEL];                  // [channel][row][34], linear in staging order
  __shared__ float s_w[9 * CCH * 16];          // [tap][channel][class]
  __shared__ float s_bias[1024];
  const int tiles_x = (p.W + CT_X - 1) / CT_X;
  const int tx0 = (blockIdx.x % tiles_x) * CT_X, ty0 = (blockIdx.x / tiles_x) * CT_Y;
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const long long HW = (long long)p.H * p.W;
  const float* xin = p.x + (long long)b * p.C * HW;

  // the folded-BatchNorm shift of every input channel, once per block
  for (int c = tid; c < p.C; c += 256) s_bias[c] = p.in_bias ? p.in_bias[c] : 0.f;

  // Staging slot `it` of a thread = element i = it*256 + tid of the linear [16][10][34] halo tile, so the LDS
  // destination is an immediate offset.  Chunk-invariant geometry is computed ONCE: the clamped source offset
  // relative to the chunk's first channel plane (22 registers), whether the element is inside the map and where the
  // channel index steps (two bit masks).  Per chunk a slot then costs one add, one load, one select, one store.
  int src_rel[NIN];
  unsigned inside_mask = 0, cstep_mask = 0;
  int c_first = 0;
  {
    int c_prev = 0;
#pragma unroll
    for (int it = 0; it < NIN; ++it) {
      const int i = min(it * 256 + tid, NEL - 1);
      const int c = i / HALO, r = i - c * HALO;
      const int ly = r / HX, lx = r - ly * HX;
      const int gy = ty0 + ly - 1, gx = tx0 + lx - 1;
      if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) inside_mask |= 1u << it;
      if (it == 0) c_first = c; else if (c != c_prev) cstep_mask |= 1u << it;
      c_prev = c;
      src_rel[it] = c * (int)HW + min(max(gy, 0), p.H - 1) * p.W + min(max(gx, 0), p.W - 1);
    }
  }
  const bool last_on = (NIN - 1) * 256 + tid < NEL;
  const int w_cls = tid & 15, w_c = tid >> 4;   // weight staging: thread -> (class, channel of the chunk), 9 taps
  const int c_last = p.C - 1;

  float pre_x[NIN], pre_w[9];
  auto load_chunk = [&](int c0) {               // global -> registers (in flight while the MFMAs of the previous chunk run)
    const float* base = xin + (long long)c0 * HW;
    const int lim = (p.C - c0) * (int)HW - 1;   // keeps reads of a ragged last chunk inside the tensor
#pragma unroll
    for (int it = 0; it < NIN; ++it) pre_x[it] = base[min(src_rel[it], lim)];
    const int wc = min(c0 + w_c, c_last), wk = min(w_cls, p.K - 1);
    const float* wp = p.w + ((long long)wk * p.C + wc) * 9;
    const bool wok = w_cls < p.K && c0 + w_c < p.C;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float v = wp[tap];
      pre_w[tap] = wok ? v : 0.f;
    }
  };
  auto store_chunk = [&](int c0) {              // registers -> LDS with the shift + ReLU fused in
    int c = c0 + c_first;
#pragma unroll
    for (int it = 0; it < NIN; ++it) {
      c += (cstep_mask >> it) & 1u;
      float v = pre_x[it] + s_bias[min(c, c_last)];
      v = p.relu ? fmaxf(v, 0.f) : v;
      v = (((inside_mask >> it) & 1u) && c < p.C) ? v : 0.f;     // zero padding outside the map / beyond C
      if (it < NIN - 1 || last_on) s_in[it * 256 + tid] = v;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) s_w[(tap * CCH + w_c) * 16 + w_cls] = pre_w[tap];
  };
  __syncthreads();                               // s_bias visible

  // wave w owns rows 2w, 2w+1 of the tile; its 4 M-tiles are (row, x-half): m -> row 2w + (m >> 1), x0 = 16 * (m & 1)
  f32x4 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_chunk(0);
  for (int c0 = 0; c0 < p.C; c0 += CCH) {
    __syncthreads();                              // every wave finished reading the previous chunk
    store_chunk(c0);
    __syncthreads();
    // The prefetch must be ISSUED here - after the LDS stores of this chunk, before its MFMAs - so that its latency hides
    // under the matrix work.  hipcc otherwise hoists the loads above the store phase, whose in-order vmcnt wait then
    // drains them immediately (measured: staging and MFMA time simply added up).
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    if (c0 + CCH < p.C) load_chunk(c0 + CCH);     // prefetch
    __builtin_amdgcn_sched_barrier(0);
    // Operand fetch is software-pipelined by hand: the 16 A + 4 B LDS reads of tap t+1 are issued before the 16 MFMAs
    // of tap t (hipcc otherwise emits read -> wait -> 2 MFMAs and exposes the LDS latency on every pair).
    float av[2][16], bv[2][4];
    auto fetch = [&](int tap, float* a, float* bq) {
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int cg = 0; cg < CCH / 4; ++cg) {
        bq[cg] = s_w[(tap * CCH + cg * 4 + g) * 16 + j];                  // B[k=g][j=class]
        const float* a0 = &s_in[(cg * 4 + g) * HALO + dx + j];            // A[i=pixel j][k=g]
#pragma unroll
        for (int m = 0; m < 4; ++m) a[cg * 4 + m] = a0[(2 * wave + (m >> 1) + dy) * HX + 16 * (m & 1)];
      }
    };
    fetch(0, av[0], bv[0]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int cur = tap & 1;
      if (tap + 1 < 9) fetch(tap + 1, av[cur ^ 1], bv[cur ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cg = 0; cg < CCH / 4; ++cg)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][cg * 4 + m], bv[cur][cg], acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // D: lane (class j, g) holds pixels 4g .. 4g+3 of each M-tile
  if (j < p.K) {
    const float bj = p.bias ? p.bias[j] : 0.f;
    float* o = p.out + ((long long)b * p.K + j) * HW;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int y = ty0 + 2 * wave + (m >> 1);
      if (y >= p.H) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = tx0 + 16 * (m & 1) + 4 * g + r;
        if (x < p.W) o[(long long)y * p.W + x] = acc[m][r] + bj;
      }
    }
  }
}

}  // namespace

extern "C" int ff3d_relu_conv3x3_small(const float* x, const float* in_bias, int apply_relu, const float* w,
                                       const float* bias, float* out, int B, int C, int H, int W, int K,
                                       ff3d_stream_t stream) {
  FF3D_REQUIRE(x && w && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && B <= 65535 && C > 0 && C <= 1024 && H > 0 && W > 0 && K > 0 && K <= 16, FF3D_ERR_BAD_SHAPE);
  ConvHeadParams p{x, in_bias, w, bias, out, C, H, W, K, apply_relu ? 1 : 0};
  const dim3 grid(((W + CT_X - 1) / CT_X) * ((H + CT_Y - 1) / CT_Y), B);
  ff3d_clear_error();
  hipLaunchKernelGGL(relu_conv3x3_small_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return ff3d_launch_status();
}
