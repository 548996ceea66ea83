// Local (k x k window) context attention for gfx950 - the MI355X counterpart of the reference's own CUDA
// extension projects/mmdet3d_plugin/models/utils/ops/locatt_ops (kernels.cuh:4-80 cc2k / ck2c_ori), used by
// LocalContextAttentionBlock (encoder_utils.py:109-163) in the `iterbev='bevfusion'` neck blocks.
//
// The reference launches one block per pixel and re-reads every key 81 times from global memory, per sample,
// materialising the (H, W, k*k) similarity tensor between two launches plus a softmax.  Here one 256-thread
// block owns an 8 x 32 pixel tile: the key (then value) halo tile of a 16-channel chunk is staged once in LDS
// (coalesced rows, zero outside the map), each thread keeps the k*k window scores of its pixel in registers,
// and similarity -> softmax -> weighting run in one launch (MODE 0) without the (H, W, k*k) round trip.
// MODE 1 / 2 expose the two reference operators separately (`similar_forward`, `weighting_forward`).
// Window positions outside the map keep a score of 0 and still take part in the softmax, exactly like the
// reference (kernels.cuh:30-40 leaves val = 0; weighting skips them, kernels.cuh:73).
// LDS-bandwidth bound (k*k LDS reads + FMAs per channel per pixel); no GEMM shape -> no MFMA.
#include <cstdlib>

#include "ff3d_common.h"

namespace {

constexpr int TY = 8, TX = 32, CC = 16;

struct LocAttParams {
  const float *q, *k, *v, *w_in;
  float *out, *w_out;
  int C, H, W;
  float scale;
};

template <int K, int MODE>
__global__ __launch_bounds__(256) void locatt_kernel(LocAttParams p) {
  constexpr int R = K / 2, HY = TY + 2 * R, HX = TX + 2 * R, PATCH = K * K;
  __shared__ float tile[CC][HY][HX + 1];
  const int tiles_x = (p.W + TX - 1) / TX;
  const int tx0 = (blockIdx.x % tiles_x) * TX, ty0 = (blockIdx.x / tiles_x) * TY;
  const int b = blockIdx.y;
  const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
  const int x = tx0 + tx, y = ty0 + ty;
  const bool valid = x < p.W && y < p.H;
  const long long HW = (long long)p.H * p.W;
  const long long img = (long long)b * p.C * HW;

  auto stage = [&](const float* src, int c0) {
    for (int i = tid; i < CC * HY * HX; i += 256) {
      const int c = i / (HY * HX), r = i - c * (HY * HX);
      const int ly = r / HX, lx = r - ly * HX;
      const int gy = ty0 + ly - R, gx = tx0 + lx - R;
      float val = 0.f;
      if (c0 + c < p.C && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) val = src[img + (c0 + c) * HW + (long long)gy * p.W + gx];
      tile[c][ly][lx] = val;
    }
  };

  float s[PATCH];
#pragma unroll
  for (int i = 0; i < PATCH; ++i) s[i] = 0.f;

  if (MODE != 2) {
    // ---- similarity: s[dy*K+dx] = sum_c q[c,y,x] * key[c,y+dy-R,x+dx-R]   (cc2k)
    for (int c0 = 0; c0 < p.C; c0 += CC) {
      __syncthreads();
      stage(p.k, c0);
      __syncthreads();
      const int cn = min(CC, p.C - c0);
      for (int c = 0; c < cn; ++c) {
        const float qv = valid ? p.q[img + (c0 + c) * HW + (long long)y * p.W + x] : 0.f;
#pragma unroll
        for (int dy = 0; dy < K; ++dy)
#pragma unroll
          for (int dx = 0; dx < K; ++dx) s[dy * K + dx] = fmaf(qv, tile[c][ty + dy][tx + dx], s[dy * K + dx]);
      }
    }
    if (MODE == 1) {
      if (valid) {
        float* o = p.w_out + ((long long)b * HW + (long long)y * p.W + x) * PATCH;
#pragma unroll
        for (int i = 0; i < PATCH; ++i) o[i] = s[i];
      }
      return;
    }
    // ---- softmax over the whole window (out-of-map positions carry a score of 0, as in the reference)
    float m = s[0] * p.scale;
#pragma unroll
    for (int i = 1; i < PATCH; ++i) m = fmaxf(m, s[i] * p.scale);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < PATCH; ++i) {
      s[i] = expf(s[i] * p.scale - m);
      sum += s[i];
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < PATCH; ++i) s[i] *= inv;
  } else if (valid) {
    const float* wi = p.w_in + ((long long)b * HW + (long long)y * p.W + x) * PATCH;
#pragma unroll
    for (int i = 0; i < PATCH; ++i) s[i] = wi[i];
  }

  // ---- weighting: out[c,y,x] = sum_k w[k] * value[c,y+dy-R,x+dx-R]   (ck2c_ori)
  for (int c0 = 0; c0 < p.C; c0 += CC) {
    __syncthreads();
    stage(p.v, c0);
    __syncthreads();
    const int cn = min(CC, p.C - c0);
    for (int c = 0; c < cn; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int dy = 0; dy < K; ++dy)
#pragma unroll
        for (int dx = 0; dx < K; ++dx) acc = fmaf(s[dy * K + dx], tile[c][ty + dy][tx + dx], acc);
      if (valid) p.out[img + (c0 + c) * HW + (long long)y * p.W + x] = acc;
    }
  }
}


// Round 5: the fused form (MODE 0) re-tiled for the LDS pipe.  The kernel above issues one 4-byte LDS read per FMA (k*k per channel
// and pixel, each lane at its own 4-byte address: 1.98 ms per call at 8 frames x 256 channels x 180 x 180, k = 9 - 12 % of the
// configs[2] step - against ~0.3 ms of fp32 FMA time).  Here a thread owns TWO horizontally adjacent pixels (x0 even): a window row
// of both is the 10 consecutive floats tile[c][y + dy][x0 .. x0 + 9], read as five 8-byte-aligned ds_read_b64 (a half-wave reads 256
// contiguous bytes: conflict-free), and serves 18 FMAs - 3.6 FMAs per LDS read instruction instead of 1.  Scores of both pixels stay
// in registers (2 x k*k), softmax per pixel, weighting the same way with one 8-byte store per channel.  Block = 256 threads = 8 rows x
// 64 pixels, 8-channel chunks (halo tile 8 x 16 x 72 fp32 = 36 KiB: two blocks per CU); the halo is staged with 16-byte loads and
// LDS writes when the window radius and the map width are multiples of 4 (k = 9 at W = 180), element-wise otherwise.
// TY2 = rows of a block's tile (threads = 32 * TY2): 8 (256 threads, two blocks per CU; the default) or 4 (128 threads, four per CU).
// Measured (round 5, sessions g / h, 8 frames x 256 channels x 180 x 180, k = 9): 1.60 ms against 1.97 ms for the one-pixel kernel;
// staging by LDS DMA instead of through registers and 4-row tiles (a finer last round) change nothing (1.66 ms): with 2 x 81 scores
// resident a wave needs the whole 256-register budget of two waves per SIMD, and at that occupancy the ~37 LDS reads per channel
// (ds_read2_b32 pairs feeding v_pk_fma_f32) are waited out one by one - LDS latency, not LDS or FMA throughput (~0.5 ms), bounds it.
constexpr int TX2 = 64, CC2 = 8;

template <int K, int TY2>
__global__ __launch_bounds__(32 * TY2, 2) void locatt2_kernel(LocAttParams p) {
  constexpr int R = K / 2, HY = TY2 + 2 * R, HX = TX2 + 2 * R, PATCH = K * K, NKV = K + 1, NT = 32 * TY2;
  static_assert(HX % 2 == 0, "8-byte aligned rows");
  __shared__ __attribute__((aligned(16))) float tile[CC2][HY][HX];
  const int tiles_x = (p.W + TX2 - 1) / TX2;
  const int tx0 = (blockIdx.x % tiles_x) * TX2, ty0 = (blockIdx.x / tiles_x) * TY2;
  const int b = blockIdx.y;
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int x0 = tx0 + 2 * tx, y = ty0 + ty;
  const bool v0 = x0 < p.W && y < p.H, v1 = x0 + 1 < p.W && y < p.H;
  const long long HW = (long long)p.H * p.W;
  const long long img = (long long)b * p.C * HW;
  const bool vec = (p.W % 4 == 0) && (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.k) | reinterpret_cast<uintptr_t>(p.v)) & 15u) == 0;

  auto stage = [&](const float* src, int c0) {
    constexpr bool VEC_OK = (R % 4 == 0) && ((CC2 * HY * (HX / 4)) % 64 == 0);      // whole waves of quads
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));       // (opaque: the per-quad address arithmetic is redone per chunk instead of living in ~20
                                        //  hoisted registers beside the 2 x k*k scores)
    if constexpr (VEC_OK) {
     if (vec) {         // rows of HX / 4 float4: gx = tx0 - R + lx is a multiple of 4 with lx, and the map edge falls between quads
      // Staged by 16-byte LDS DMA (no registers: the 2 x 81 scores fill the budget; the quad-by-quad loop through registers left the
      // kernel at 1.60 ms - 64 stagings per block of ~9 dependent global round trips each, session g).  The tile is lane-linear in
      // quads (quad i = (c, ly, lx / 4) in memory order), so wave w's 64 lanes of pass `it` land at quad (it * 256 + w * 64); quads
      // outside the map are fetched from a clamped address and overwritten with zeros once the DMAs have landed.
      constexpr int Q = HX / 4, TOTAL = CC2 * HY * Q, NV = (TOTAL + NT - 1) / NT;      // (last pass: only the waves below TOTAL)
      static_assert(NV <= 32, "out_of_map bit mask");
      float* const t0 = &tile[0][0][0];
      unsigned out_of_map = 0;
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        const int i = it * NT + tid;
        if (it * NT + (tid & ~63) >= TOTAL) continue;                     // wave-uniform
        const int c = i / (HY * Q), r = i - c * (HY * Q);
        const int ly = r / Q, lx = (r - ly * Q) * 4;
        const int gy = ty0 + ly - R, gx = tx0 + lx - R;
        const bool ok = c0 + c < p.C && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const float* a = ok ? src + img + (c0 + c) * HW + (long long)gy * p.W + gx : src;
        if (!ok) out_of_map |= 1u << it;
        __builtin_amdgcn_global_load_lds(a, (__attribute__((address_space(3))) void*)(t0 + (it * NT + (tid & ~63)) * 4), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < NV; ++it)
        if (out_of_map >> it & 1u) reinterpret_cast<float4*>(t0)[it * NT + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
      return;
     }
    }
    {
      for (int i = tid; i < CC2 * HY * HX; i += NT) {
        const int c = i / (HY * HX), r = i - c * (HY * HX);
        const int ly = r / HX, lx = r - ly * HX;
        const int gy = ty0 + ly - R, gx = tx0 + lx - R;
        float val = 0.f;
        if (c0 + c < p.C && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) val = src[img + (c0 + c) * HW + (long long)gy * p.W + gx];
        tile[c][ly][lx] = val;
      }
    }
  };
  // the K + 1 floats both pixels' window row dy needs: tile[c][ty + dy][2 tx .. 2 tx + K] (8-byte aligned: 2 tx and HX are even)
  auto row = [&](int c, int dy, float (&kv)[NKV + 1]) {
    const float2* src = reinterpret_cast<const float2*>(&tile[c][ty + dy][2 * tx]);
#pragma unroll
    for (int j = 0; j < (NKV + 1) / 2; ++j) {
      const float2 t = src[j];
      kv[2 * j] = t.x, kv[2 * j + 1] = t.y;
    }
  };

  float s0[PATCH], s1[PATCH];
#pragma unroll
  for (int i = 0; i < PATCH; ++i) s0[i] = 0.f, s1[i] = 0.f;

  // ---- similarity (cc2k): s[dy * K + dx] = sum_c q[c, y, x] * key[c, y + dy - R, x + dx - R]
  for (int c0 = 0; c0 < p.C; c0 += CC2) {
    __syncthreads();
    stage(p.k, c0);
    const int cn = min(CC2, p.C - c0);
    // the two query values of a channel are fetched one channel ahead (two registers in flight instead of 2 x CC2: with 2 x 81
    // scores resident the kernel sits at the 256-register budget of two blocks per CU)
    const float* qrow = p.q + img + (long long)min(y, p.H - 1) * p.W;
    auto load_q = [&](int c, float& a, float& b2) {
      const float* qp = qrow + (long long)min(c0 + c, p.C - 1) * HW;
      a = v0 ? qp[x0] : 0.f;
      b2 = v1 ? qp[x0 + 1] : 0.f;
    };
    float qn0, qn1;
    load_q(0, qn0, qn1);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CC2; ++c) {
      if (c >= cn) break;
      const float q0 = qn0, q1 = qn1;
      if (c + 1 < cn) load_q(c + 1, qn0, qn1);
#pragma unroll
      for (int dy = 0; dy < K; ++dy) {
        float kv[NKV + 1];
        row(c, dy, kv);
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
          s0[dy * K + dx] = fmaf(q0, kv[dx], s0[dy * K + dx]);
          s1[dy * K + dx] = fmaf(q1, kv[dx + 1], s1[dy * K + dx]);
        }
      }
    }
  }
  // ---- softmax over the whole window, per pixel (out-of-map positions carry a score of 0, as in the reference)
  auto softmax = [&](float (&s)[PATCH]) {
    float m = s[0] * p.scale;
#pragma unroll
    for (int i = 1; i < PATCH; ++i) m = fmaxf(m, s[i] * p.scale);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < PATCH; ++i) {
      s[i] = expf(s[i] * p.scale - m);
      sum += s[i];
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < PATCH; ++i) s[i] *= inv;
  };
  softmax(s0);
  softmax(s1);

  // ---- weighting (ck2c_ori): out[c, y, x] = sum_k w[k] * value[c, y + dy - R, x + dx - R]
  const bool pair_store = (p.W % 2 == 0) && (reinterpret_cast<uintptr_t>(p.out) & 7u) == 0 && (HW % 2 == 0);
  for (int c0 = 0; c0 < p.C; c0 += CC2) {
    __syncthreads();
    stage(p.v, c0);
    __syncthreads();
    const int cn = min(CC2, p.C - c0);
#pragma unroll
    for (int c = 0; c < CC2; ++c) {
      if (c >= cn) break;
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int dy = 0; dy < K; ++dy) {
        float kv[NKV + 1];
        row(c, dy, kv);
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
          a0 = fmaf(s0[dy * K + dx], kv[dx], a0);
          a1 = fmaf(s1[dy * K + dx], kv[dx + 1], a1);
        }
      }
      float* o = p.out + img + (c0 + c) * HW + (long long)y * p.W + x0;
      if (v1 && pair_store) {
        *reinterpret_cast<float2*>(o) = make_float2(a0, a1);
      } else {
        if (v0) o[0] = a0;
        if (v1) o[1] = a1;
      }
    }
  }
}

// ck2c_loc (kernels.cuh:82-119): y[c, h, w] = sum_k x[c, h - dy, w - dx] * weight[(h - dy, w - dx), k], (dy, dx) = k's offset
// from the window centre - the transpose of ck2c_ori, i.e. the gradient of `similar` with respect to its second operand
// (x = x_ori, weight = grad) and of `weighting` with respect to its first (x = grad, weight = x_weight).  Same 8 x 32 tile:
// the x halo tile of a 16-channel chunk is staged in LDS; the K*K weights a pixel needs sit at K*K DIFFERENT source pixels
// (one entry of each neighbour's window), so they are gathered once per pixel into registers before the channel loop.
template <int K>
__global__ __launch_bounds__(256) void locatt_loc_kernel(LocAttParams p) {
  constexpr int R = K / 2, HY = TY + 2 * R, HX = TX + 2 * R, PATCH = K * K;
  __shared__ float tile[CC][HY][HX + 1];
  const int tiles_x = (p.W + TX - 1) / TX;
  const int tx0 = (blockIdx.x % tiles_x) * TX, ty0 = (blockIdx.x / tiles_x) * TY;
  const int b = blockIdx.y;
  const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
  const int x = tx0 + tx, y = ty0 + ty;
  const bool valid = x < p.W && y < p.H;
  const long long HW = (long long)p.H * p.W;
  const long long img = (long long)b * p.C * HW;
  // weight of source pixel (y - dy + R, x - dx + R) for window entry (dy, dx); 0 where the source is outside the map
  float s[PATCH];
#pragma unroll
  for (int dy = 0; dy < K; ++dy)
#pragma unroll
    for (int dx = 0; dx < K; ++dx) {
      const int sy = y + R - dy, sx = x + R - dx;
      s[dy * K + dx] = (valid && sy >= 0 && sy < p.H && sx >= 0 && sx < p.W)
                           ? p.w_in[((long long)b * HW + (long long)sy * p.W + sx) * PATCH + dy * K + dx] : 0.f;
    }
  for (int c0 = 0; c0 < p.C; c0 += CC) {
    __syncthreads();
    for (int i = tid; i < CC * HY * HX; i += 256) {
      const int c = i / (HY * HX), r = i - c * (HY * HX);
      const int ly = r / HX, lx = r - ly * HX;
      const int gy = ty0 + ly - R, gx = tx0 + lx - R;
      float val = 0.f;
      if (c0 + c < p.C && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) val = p.v[img + (c0 + c) * HW + (long long)gy * p.W + gx];
      tile[c][ly][lx] = val;
    }
    __syncthreads();
    const int cn = min(CC, p.C - c0);
    for (int c = 0; c < cn; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int dy = 0; dy < K; ++dy)
#pragma unroll
        for (int dx = 0; dx < K; ++dx) acc = fmaf(s[dy * K + dx], tile[c][ty + 2 * R - dy][tx + 2 * R - dx], acc);
      if (valid) p.out[img + (c0 + c) * HW + (long long)y * p.W + x] = acc;
    }
  }
}

// the fused form on the two-pixel kernel (FF3D_LOCATT_V2=0: the one-pixel kernel of rounds 1-4, A/B runs)
template <int TY2>
int launch_fused_ty(int K, const LocAttParams& p, int B, hipStream_t s) {
  const dim3 grid(((p.W + TX2 - 1) / TX2) * ((p.H + TY2 - 1) / TY2), B), block(32 * TY2);
  ff3d_clear_error();
  switch (K) {
    case 1: hipLaunchKernelGGL((locatt2_kernel<1, TY2>), grid, block, 0, s, p); break;
    case 3: hipLaunchKernelGGL((locatt2_kernel<3, TY2>), grid, block, 0, s, p); break;
    case 5: hipLaunchKernelGGL((locatt2_kernel<5, TY2>), grid, block, 0, s, p); break;
    case 7: hipLaunchKernelGGL((locatt2_kernel<7, TY2>), grid, block, 0, s, p); break;
    case 9: hipLaunchKernelGGL((locatt2_kernel<9, TY2>), grid, block, 0, s, p); break;
    default: return FF3D_ERR_UNSUPPORTED;
  }
  return ff3d_launch_status();
}

int launch_fused(int K, const LocAttParams& p, int B, hipStream_t s) {
  // 8-row tiles; FF3D_LOCATT_TY=4 selects the 4-row tiles (A/B record: 1.66 vs 1.60 ms at 8 frames of 180 x 180 - the finer
  // grid does not pay, the kernel waits on LDS latency at two waves per SIMD, not on its last round)
  static const int ty_force = [] {
    const char* e = getenv("FF3D_LOCATT_TY");
    return e ? atoi(e) : 0;
  }();
  return ty_force == 4 ? launch_fused_ty<4>(K, p, B, s) : launch_fused_ty<8>(K, p, B, s);
}

template <int MODE>
int launch(int K, const LocAttParams& p, int B, hipStream_t s) {
  const dim3 grid(((p.W + TX - 1) / TX) * ((p.H + TY - 1) / TY), B), block(256);
  ff3d_clear_error();
  switch (K) {
    case 1: hipLaunchKernelGGL((locatt_kernel<1, MODE>), grid, block, 0, s, p); break;
    case 3: hipLaunchKernelGGL((locatt_kernel<3, MODE>), grid, block, 0, s, p); break;
    case 5: hipLaunchKernelGGL((locatt_kernel<5, MODE>), grid, block, 0, s, p); break;
    case 7: hipLaunchKernelGGL((locatt_kernel<7, MODE>), grid, block, 0, s, p); break;
    case 9: hipLaunchKernelGGL((locatt_kernel<9, MODE>), grid, block, 0, s, p); break;
    default: return FF3D_ERR_UNSUPPORTED;
  }
  return ff3d_launch_status();
}

bool shape_ok(int B, int C, int H, int W, int kH, int kW) {
  return B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && kH == kW && (kH & 1) && kH <= 9;
}

}  // namespace

extern "C" int ff3d_locatt_similar(const float* x_ori, const float* x_loc, float* y, int B, int C, int H, int W, int kH,
                                   int kW, ff3d_stream_t stream) {
  FF3D_REQUIRE(x_ori && x_loc && y, FF3D_ERR_NULL);
  FF3D_REQUIRE(shape_ok(B, C, H, W, kH, kW), FF3D_ERR_BAD_SHAPE);
  LocAttParams p{x_ori, x_loc, nullptr, nullptr, nullptr, y, C, H, W, 1.f};
  return launch<1>(kH, p, B, static_cast<hipStream_t>(stream));
}

extern "C" int ff3d_locatt_weighting(const float* x_ori, const float* x_weight, float* y, int B, int C, int H, int W,
                                     int kH, int kW, ff3d_stream_t stream) {
  FF3D_REQUIRE(x_ori && x_weight && y, FF3D_ERR_NULL);
  FF3D_REQUIRE(shape_ok(B, C, H, W, kH, kW), FF3D_ERR_BAD_SHAPE);
  LocAttParams p{nullptr, nullptr, x_ori, x_weight, y, nullptr, C, H, W, 1.f};
  return launch<2>(kH, p, B, static_cast<hipStream_t>(stream));
}

extern "C" int ff3d_locatt_ck2c_loc(const float* x, const float* weight, float* y, int B, int C, int H, int W, int kH, int kW,
                                    ff3d_stream_t stream) {
  FF3D_REQUIRE(x && weight && y, FF3D_ERR_NULL);
  FF3D_REQUIRE(shape_ok(B, C, H, W, kH, kW), FF3D_ERR_BAD_SHAPE);
  LocAttParams p{nullptr, nullptr, x, weight, y, nullptr, C, H, W, 1.f};
  const dim3 grid(((W + TX - 1) / TX) * ((H + TY - 1) / TY), B), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  ff3d_clear_error();
  switch (kH) {
    case 1: hipLaunchKernelGGL(locatt_loc_kernel<1>, grid, block, 0, s, p); break;
    case 3: hipLaunchKernelGGL(locatt_loc_kernel<3>, grid, block, 0, s, p); break;
    case 5: hipLaunchKernelGGL(locatt_loc_kernel<5>, grid, block, 0, s, p); break;
    case 7: hipLaunchKernelGGL(locatt_loc_kernel<7>, grid, block, 0, s, p); break;
    case 9: hipLaunchKernelGGL(locatt_loc_kernel<9>, grid, block, 0, s, p); break;
    default: return FF3D_ERR_UNSUPPORTED;
  }
  return ff3d_launch_status();
}

extern "C" int ff3d_local_attention(const float* query, const float* key, const float* value, float* out, int B, int C,
                                    int H, int W, int kH, int kW, float scale, ff3d_stream_t stream) {
  FF3D_REQUIRE(query && key && value && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(shape_ok(B, C, H, W, kH, kW), FF3D_ERR_BAD_SHAPE);
  LocAttParams p{query, key, value, nullptr, out, nullptr, C, H, W, scale};
  static const bool v2 = [] {
    const char* e = getenv("FF3D_LOCATT_V2");
    return !(e && e[0] == '0');
  }();
  if (v2) return launch_fused(kH, p, B, static_cast<hipStream_t>(stream));
  return launch<0>(kH, p, B, static_cast<hipStream_t>(stream));
}
