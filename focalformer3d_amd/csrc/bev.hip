// BEV pyramid flatten (NCHW levels -> one channels-last (B, Nv, C) tensor, + positional embedding),
// generic NCHW->NHWC transpose, and the sine positional embedding.  Pure HBM-bound data movement:
// 64x64 LDS-tiled transposes with 256-byte coalesced rows on both the read and the write side.
#include <cstdlib>
#include <cstring>

#include "ff3d_common.h"

namespace {

constexpr int TT = 64;  // transpose tile edge

struct FlattenParams {
  const float* level[FF3D_MAX_LEVELS];
  int tile_start[FF3D_MAX_LEVELS + 1];  // cumulative number of n-tiles per level
  LevelTable lv;
  // up to FL_MAXV value tensors (one per decoder stage: value_s = pyramid + pos_embed_s) written by the same pass
  const float* pos_embed[4];
  float* out_raw;
  float* out_value[4];         // fp32 (B, Nv, C), or with value_split two fp16 planes (hi, lo') of that shape
  int n_values;
  int value_split;             // value_dtype: 0 fp32, FF3D_BF16 one bf16 plane (round 5), FF3D_F16_SPLIT the (hi, lo') pair
  long long value_plane;       // halves per plane
  int C, B, c_tiles, frame_fastest;
  int fb;                      // frames a block walks (round 5: the positional tile is read once per block, not once per frame)
  int vec4;   // C % 4 == 0 and every base pointer 16-byte aligned
  // range normalisation of the split value (ff3d.h): bound exponents of the inputs, exponents written for the outputs
  const int* level_exp[FF3D_MAX_LEVELS];
  const int* pe_exp[4];
  int* value_exp[4];
  int* raw_exp;
  int scaled;
};

// in: (C, HW) plane set of one batch element; out rows (n, C).
__device__ __forceinline__ void transpose_tile(const float* __restrict__ in, long long in_c_stride, int HW, int C,
                                               int n0, int c0, const float* __restrict__ pe, float* __restrict__ o1,
                                               float* __restrict__ o2, float (*tile)[TT + 1], long long split_plane = 0,
                                               float split_scale = 1.f, bool load = true) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  if (load) {                                              // (further write phases reuse the tile already in LDS)
    for (int r = ty; r < TT; r += 4) {
      const int c = c0 + r, n = n0 + tx;
      tile[r][tx] = (c < C && n < HW) ? in[(long long)c * in_c_stride + n] : 0.f;
    }
    __syncthreads();
  }
  for (int r = ty; r < TT; r += 4) {
    const int n = n0 + r, c = c0 + tx;
    if (n < HW && c < C) {
      const float v = tile[tx][r];
      const long long o = (long long)n * C + c;
      if (o1) o1[o] = v;
      if (o2) {
        const float w = pe ? v + pe[o] : v;
        if (split_plane < 0) {                               // one bf16 plane (round to nearest even)
          reinterpret_cast<__bf16*>(o2)[o] = (__bf16)w;
        } else if (split_plane) {
          _Float16* h = reinterpret_cast<_Float16*>(o2);
          const float ws = w * split_scale;
          const _Float16 hi = (_Float16)ws;
          h[o] = hi;
          h[split_plane + o] = (_Float16)((ws - (float)hi) * 2048.f);
        } else {
          o2[o] = w;
        }
      }
    }
  }
}

// Same tile with 16-byte global accesses on both sides (needs HW % 4 == 0, C % 4 == 0 and 16-byte aligned bases):
// read phase = float4 along n (16 lanes cover one channel's 64 cells), write phase = float4 along c (16 lanes cover
// one cell's 64 channels); LDS accesses stay scalar with the 65-float row stride (<= 2-way conflicts).
__device__ __forceinline__ void transpose_tile_v4(const float* __restrict__ in, long long in_c_stride, int HW, int C,
                                                  int n0, int c0, const float* __restrict__ pe, float* __restrict__ o1,
                                                  float* __restrict__ o2, float (*tile)[TT + 1], long long split_plane = 0,
                                                  float split_scale = 1.f, bool load = true, const float4* pe_reg = nullptr) {
  const int l16 = threadIdx.x & 15, r16 = threadIdx.x >> 4;   // 16 x 16
  if (load) {
    for (int r = r16; r < TT; r += 16) {
      const int c = c0 + r, n = n0 + 4 * l16;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < C && n < HW) v = *reinterpret_cast<const float4*>(in + (long long)c * in_c_stride + n);   // HW % 4 == 0
      tile[r][4 * l16 + 0] = v.x;
      tile[r][4 * l16 + 1] = v.y;
      tile[r][4 * l16 + 2] = v.z;
      tile[r][4 * l16 + 3] = v.w;
    }
    __syncthreads();
  }
#pragma unroll
  for (int it = 0; it < TT / 16; ++it) {
    const int r = r16 + 16 * it;
    const int n = n0 + r, c = c0 + 4 * l16;
    if (n < HW && c < C) {
      float4 v = make_float4(tile[4 * l16][r], tile[4 * l16 + 1][r], tile[4 * l16 + 2][r], tile[4 * l16 + 3][r]);
      const long long o = (long long)n * C + c;
      if (o1) *reinterpret_cast<float4*>(o1 + o) = v;
      if (o2) {
        if (pe_reg) {                                        // the block's positional tile, read once for all of its frames
          const float4 q = pe_reg[it];
          v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        } else if (pe) {
          const float4 q = *reinterpret_cast<const float4*>(pe + o);
          v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        if (split_plane < 0) {                               // one bf16 plane: 4 channels = one 8-byte store
          __bf16 q[4] = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
          *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(o2) + o) = *reinterpret_cast<uint2*>(q);
        } else if (split_plane) {
          _Float16* h = reinterpret_cast<_Float16*>(o2);
          const float f[4] = {v.x * split_scale, v.y * split_scale, v.z * split_scale, v.w * split_scale};
          _Float16 hi[4], lo[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            hi[k] = (_Float16)f[k];
            lo[k] = (_Float16)((f[k] - (float)hi[k]) * 2048.f);
          }
          *reinterpret_cast<uint2*>(h + o) = *reinterpret_cast<uint2*>(hi);
          *reinterpret_cast<uint2*>(h + split_plane + o) = *reinterpret_cast<uint2*>(lo);
        } else {
          *reinterpret_cast<float4*>(o2 + o) = v;
        }
      }
    }
  }
}

// Grid: one dimension, XCD-remapped.  Round 5 measured a FRAME-FASTEST order (the B blocks that transpose the same (pixel tile,
// channel tile) of different frames back to back on one XCD, so that the positional-embedding tile they all add - Nv x C fp32 per
// decoder stage: 43.5 MB at 180 x 180, 294 MB at 468 x 468 - comes from HBM once and from that XCD's L2 for the other frames):
// SLOWER, 1.82 vs 1.67 ms at 468 x 468 x 8 frames, 1.32 vs 1.27 ms at 180 x 180 x 32 (profiles/r05_f_*): the table reads were
// cache hits already, and eight blocks writing eight frames' far-apart output rows at once cost more than they save.  The
// frame-slowest order of rounds 1-4 stays the default; FF3D_FLATTEN_ORDER=frame-fastest selects the other (A/B record).
__global__ __launch_bounds__(256) void bev_flatten_kernel(FlattenParams p) {
  __shared__ float tile[TT][TT + 1];
  const unsigned lid = ff3d_xcd_remap(blockIdx.x, gridDim.x);
  const unsigned groups = (unsigned)((p.B + p.fb - 1) / p.fb), per_group = gridDim.x / groups;
  const int grp = p.frame_fastest ? (int)(lid % groups) : (int)(lid / per_group);
  const unsigned rest = p.frame_fastest ? lid / groups : lid % per_group;
  const int ct = (int)(rest % (unsigned)p.c_tiles), tl = (int)(rest / (unsigned)p.c_tiles);
  int l = 0;
  while (l + 1 < p.lv.L && tl >= p.tile_start[l + 1]) ++l;
  const int HW = p.lv.H[l] * p.lv.W[l];
  const int n0 = (tl - p.tile_start[l]) * TT, c0 = ct * TT;
  const long long plane = p.value_split == FF3D_F16_SPLIT ? p.value_plane : (p.value_split == FF3D_BF16 ? -1 : 0);
  const bool first_block = lid == 0 && threadIdx.x == 0;
  int e_raw = 0;
  if (p.scaled) {
    e_raw = ff3d_ld_exp(p.level_exp[0]);
    for (int k = 1; k < p.lv.L; ++k) e_raw = max(e_raw, ff3d_ld_exp(p.level_exp[k]));
    if (first_block && p.raw_exp) *p.raw_exp = e_raw;
  }
  const bool v4 = p.vec4 && (HW & 3) == 0;
  // the tile is read from HBM and transposed through LDS once; every value tensor is one more write phase over it
  const int passes = max(p.n_values, 1);
  // Round 5: a block walks `fb` frames of its (pixel tile, channel tile) and keeps the positional tiles of the value tensors in
  // registers (4 float4 per value and thread in the 16-byte path).  Before, every frame's block read them again: Nv x C fp32 per
  // decoder stage - 2 x 43.5 MB at 180 x 180, 2 x 294 MB at 468 x 468, neither resident in an L2 across frames - 2.8 GB of the
  // 4.1 GB the kernel fetched per 32-frame launch (PMC FETCH_SIZE, profiles/r05_l_*), 4.7 of 7.1 GB at 468 x 468 x 8 frames.
  const bool pe_regs = v4 && p.fb > 1;
  float4 per0[TT / 16], per1[TT / 16], per2[TT / 16], per3[TT / 16];     // (four named arrays: indexed statically -> registers)
  auto load_pe = [&](int v, float4* pr) {
    const int l16 = threadIdx.x & 15, r16 = threadIdx.x >> 4;
#pragma unroll
    for (int it = 0; it < TT / 16; ++it) {
      const int n = n0 + r16 + 16 * it, c = c0 + 4 * l16;
      pr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pe_regs && v < p.n_values && p.pos_embed[v] && n < HW && c < p.C)
        pr[it] = *reinterpret_cast<const float4*>(p.pos_embed[v] + ((long long)p.lv.start[l] + n) * p.C + c);
    }
  };
  load_pe(0, per0), load_pe(1, per1), load_pe(2, per2), load_pe(3, per3);
  for (int bb = 0; bb < p.fb; ++bb) {
    const int b = grp * p.fb + bb;
    if (b >= p.B) break;
    if (bb) __syncthreads();                    // every thread is done with the previous frame's tile
    const float* in = p.level[l] + (long long)b * p.C * HW;
    const long long row0 = (long long)b * p.lv.Nv + p.lv.start[l];
    float* o1 = p.out_raw ? p.out_raw + row0 * p.C : nullptr;
    auto one = [&](int v, const float4* pr) {
      const bool has = v < p.n_values;
      const float* pe = (has && p.pos_embed[v]) ? p.pos_embed[v] + (long long)p.lv.start[l] * p.C : nullptr;
      // split / bf16 value: o2 addresses 2-byte elements, so the row offset is applied in halves
      float* o2 = !has ? nullptr
                  : p.value_split ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(p.out_value[v]) + row0 * p.C)
                                  : p.out_value[v] + row0 * p.C;
      // value = level + pos_embed: |value| < 2^(max_l e_l + 15) + 2^(e_pe + 15) <= 2^(max(e_l, e_pe) + 16)
      float split_scale = 1.f;
      if (p.scaled && has) {
        const int e_val = ((has && p.pos_embed[v]) ? max(e_raw, ff3d_ld_exp(p.pe_exp[v])) : e_raw - 1) + 1;
        split_scale = ff3d_pow2(-e_val);
        if (first_block && bb == 0 && p.value_exp[v]) *p.value_exp[v] = e_val;
      }
      if (v4)
        transpose_tile_v4(in, HW, HW, p.C, n0, c0, pe, v == 0 ? o1 : nullptr, o2, tile, plane, split_scale, v == 0,
                          (pe_regs && pe) ? pr : nullptr);
      else
        transpose_tile(in, HW, HW, p.C, n0, c0, pe, v == 0 ? o1 : nullptr, o2, tile, plane, split_scale, v == 0);
    };
    one(0, per0);
    if (passes > 1) one(1, per1);
    if (passes > 2) one(2, per2);
    if (passes > 3) one(3, per3);
  }
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int C, int HW, int vec4) {
  __shared__ float tile[TT][TT + 1];
  const int n0 = blockIdx.x * TT, c0 = blockIdx.y * TT;
  const long long img = blockIdx.z;
  if (vec4)
    transpose_tile_v4(in + img * C * HW, HW, HW, C, n0, c0, nullptr, out + img * C * HW, nullptr, tile);
  else
    transpose_tile(in + img * C * HW, HW, HW, C, n0, c0, nullptr, out + img * C * HW, nullptr, tile);
}

// emb[n, i]: i < 128 -> y embedding, i >= 128 -> x embedding (UT:53 cat((pos_y, pos_x))).
__global__ __launch_bounds__(256) void sine_embed_kernel(const float* __restrict__ pos,
                                                         const float* __restrict__ dim_t, float* __restrict__ emb,
                                                         long long N, float W, float H) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= N * 256) return;
  const long long n = e >> 8;
  const int i = (int)(e & 255), j = i & 127;
  const float scale = 6.283185307179586f;
  const float r = (i < 128) ? pos[n * 2 + 1] / H : pos[n * 2] / W;  // FD:869 reference = pos / (W, H)
  const float v = (r * scale) / dim_t[j];
  emb[e] = (j & 1) ? cosf(v) : sinf(v);
}

}  // namespace

extern "C" int ff3d_bev_flatten_multi(const float* const* levels_host, int n_values, const float* const* pos_embeds_host,
                                      float* out_raw, void* const* out_values_host, int value_dtype, int B, int C, int L,
                                      const int32_t* level_hw_host, const int32_t* const* level_exp_host,
                                      const int32_t* const* pe_exps_host, int32_t* const* value_exps_host, int32_t* raw_exp,
                                      ff3d_stream_t stream) {
  FF3D_REQUIRE(levels_host && (out_raw || n_values > 0), FF3D_ERR_NULL);
  FF3D_REQUIRE(n_values >= 0 && n_values <= 4 && (n_values == 0 || out_values_host), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(value_dtype == FF3D_F32 || value_dtype == FF3D_F16_SPLIT || value_dtype == FF3D_BF16, FF3D_ERR_BAD_DTYPE);
  FF3D_REQUIRE(value_dtype != FF3D_BF16 || !level_exp_host, FF3D_ERR_BAD_DTYPE);     // (the bf16 plane carries no exponent)
  FF3D_REQUIRE(B > 0 && B <= 65535 && C > 0, FF3D_ERR_BAD_SHAPE);
  FlattenParams p;
  FF3D_REQUIRE(ff3d_make_levels(level_hw_host, L, &p.lv), FF3D_ERR_BAD_SHAPE);
  int tiles = 0;
  for (int l = 0; l < FF3D_MAX_LEVELS; ++l) {
    p.level[l] = l < L ? levels_host[l] : nullptr;
    p.tile_start[l] = tiles;
    if (l < L) {
      FF3D_REQUIRE(levels_host[l], FF3D_ERR_NULL);
      tiles += (p.lv.H[l] * p.lv.W[l] + TT - 1) / TT;
    }
  }
  p.tile_start[FF3D_MAX_LEVELS] = tiles;
  p.scaled = level_exp_host != nullptr;
  for (int l = 0; l < FF3D_MAX_LEVELS; ++l) p.level_exp[l] = (level_exp_host && l < L) ? level_exp_host[l] : nullptr;
  p.raw_exp = raw_exp;
  p.out_raw = out_raw;
  p.n_values = n_values;
  p.value_split = value_dtype;
  p.value_plane = ((long long)B * p.lv.Nv + 1) * C;   // + the zero row of the split-GEMM operand contract
  p.C = C, p.B = B, p.c_tiles = (C + TT - 1) / TT;
  static const bool frame_fastest = [] {
    const char* e = getenv("FF3D_FLATTEN_ORDER");
    return e && strncmp(e, "frame-f", 7) == 0;      // "frame-fastest"
  }();
  p.frame_fastest = frame_fastest ? 1 : 0;
  static const int fb_env = [] {                    // frames per block; FF3D_FLATTEN_FB=1: one block per frame (rounds 1-4)
    const char* e = getenv("FF3D_FLATTEN_FB");
    return e ? atoi(e) : 0;
  }();
  // default: the largest of 8 / 4 / 2 / 1 frames per block that still leaves >= 8 192 blocks (32 per CU): 32 frames at 180 x 180 -> 8
  // (1295 vs 1271 frames/s), 4 frames -> 1 (4 per block: 1166 vs 1181, profiles/r05_aj_*), 8 frames at 468 x 468 -> 8
  int fb = fb_env;
  if (fb < 1)
    for (fb = 8; fb > 1; fb >>= 1)
      if (fb <= B && (long long)tiles * p.c_tiles * ((B + fb - 1) / fb) >= 8192) break;
  p.fb = (n_values > 0 && pos_embeds_host) ? (fb > B ? B : fb) : 1;
  const int groups = (B + p.fb - 1) / p.fb;
  FF3D_REQUIRE((long long)tiles * p.c_tiles * groups < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  p.vec4 = (C % 4 == 0) && ff3d_aligned16(out_raw);
  for (int v = 0; v < 4; ++v) {
    const bool has = v < n_values;
    p.pos_embed[v] = (has && pos_embeds_host) ? pos_embeds_host[v] : nullptr;
    p.out_value[v] = has ? static_cast<float*>(out_values_host[v]) : nullptr;
    p.pe_exp[v] = (has && pe_exps_host) ? pe_exps_host[v] : nullptr;
    p.value_exp[v] = (has && value_exps_host) ? value_exps_host[v] : nullptr;
    if (has) {
      FF3D_REQUIRE(p.out_value[v], FF3D_ERR_NULL);
      FF3D_REQUIRE(!p.scaled || !p.pos_embed[v] || p.pe_exp[v], FF3D_ERR_NULL);
      p.vec4 = p.vec4 && ff3d_aligned16(p.pos_embed[v]) && ff3d_aligned16(p.out_value[v]);
    }
  }
  for (int l = 0; l < L; ++l) p.vec4 = p.vec4 && ff3d_aligned16(levels_host[l]);
  ff3d_clear_error();
  hipLaunchKernelGGL(bev_flatten_kernel, dim3((unsigned)(tiles * p.c_tiles * groups)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  return ff3d_launch_status();
}

extern "C" int ff3d_bev_flatten(const float* const* levels_host, const float* pos_embed, float* out_raw,
                                void* out_value, int value_dtype, int B, int C, int L, const int32_t* level_hw_host,
                                const int32_t* const* level_exp_host, const int32_t* pe_exp, int32_t* value_exp,
                                int32_t* raw_exp, ff3d_stream_t stream) {
  FF3D_REQUIRE(levels_host && (out_raw || out_value), FF3D_ERR_NULL);
  const float* pes[1] = {pos_embed};
  void* outs[1] = {out_value};
  const int32_t* pexp[1] = {pe_exp};
  int32_t* vexp[1] = {value_exp};
  return ff3d_bev_flatten_multi(levels_host, out_value ? 1 : 0, pes, out_raw, outs, value_dtype, B, C, L, level_hw_host,
                                level_exp_host, pexp, vexp, raw_exp, stream);
}

extern "C" int ff3d_nchw_to_nhwc(const float* in, float* out, int N, int C, int HW, ff3d_stream_t stream) {
  FF3D_REQUIRE(in && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(N > 0 && N <= 65535 && C > 0 && HW > 0, FF3D_ERR_BAD_SHAPE);
  ff3d_clear_error();
  const int vec4 = (C % 4 == 0) && (HW % 4 == 0) && ff3d_aligned16(in) && ff3d_aligned16(out);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((HW + TT - 1) / TT, (C + TT - 1) / TT, N), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, out, C, HW, vec4);
  return ff3d_launch_status();
}

extern "C" int ff3d_sine_embed(const float* pos, const float* dim_t, float* emb, int64_t N, float W, float H,
                               ff3d_stream_t stream) {
  FF3D_REQUIRE(pos && dim_t && emb, FF3D_ERR_NULL);
  FF3D_REQUIRE(N > 0 && W > 0.f && H > 0.f, FF3D_ERR_BAD_SHAPE);
  const long long blocks = (N * 256 + 255) / 256;
  FF3D_REQUIRE(blocks < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  ff3d_clear_error();
  hipLaunchKernelGGL(sine_embed_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), pos,
                     dim_t, emb, (long long)N, W, H);
  return ff3d_launch_status();
}
