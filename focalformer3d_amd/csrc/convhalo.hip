// 3x3 convolution (stride 1, padding 1) with fp32-class accuracy on the fp16 matrix cores, halo-tile form.
//
// Same arithmetic as splitmm.hip - operands as (hi, lo') fp16 pairs, three v_mfma_f32_16x16x32_f16 passes per product,
// fp32 accumulation - but a different data flow.  The implicit GEMM of splitmm.hip stages every activation nine times (once
// per filter tap) and its loop is bound by the ISSUE cost of the LDS DMA (8 one-KiB pieces per wave per K-step ~ the 768
// cycles of that step's 48 MFMAs, measured: 56 % MFMA-busy whatever the tile / pipeline depth).  Here a 512-thread block
// owns a 4 x 64 pixel tile x 128 output channels and per 32-channel chunk DMAs the 6 x 66 halo ONCE (49.5 KiB for both
// planes, double-buffered across chunks) and serves all nine taps from LDS; only the weights are streamed per tap (16 KiB,
// double-buffered).  DMA pieces per wave per (chunk, tap) step: 2 (weights) + <= 1 (next halo) instead of 8.
//   waves: 8 = 4 (pixel rows) x 2 (64-channel halves); wave tile 64 pixels x 64 channels = 4 x 4 MFMA tiles x 2 accumulators
//   LDS:   halo [2][plane][396 px][32 ch] 99 KiB + weights [2][plane][128 n][32 ch] 32 KiB = 131 KiB, one block per CU
//   LDS rows are 64 B with the XOR chunk swizzle of splitmm.hip (on the DMA source address and on the fragment read)
// (Tried, same-box A/B: weights two taps ahead through three buffers with counted s_waitcnt vmcnt + raw s_barrier - 11 % slower
// than this vmcnt(0) + __syncthreads form: 3.27 vs 2.95 ms; s_setprio(1) around the MFMA block 3.68 ms; iglp_opt(0) 3.90 ms.
// Round 2: a two-group ping-pong (4 waves fetch fragments while the other 4 issue MFMAs, two barriers per step) 3.19 vs 3.01 ms;
// leaving the next chunk's halo DMAs in flight across one more step (vmcnt(2)) 3.25 vs 2.90 ms.)
// Where the time goes (FF3D_HALO_ABLATE, B=32, 256 -> 256, 180 x 180; ms): everything 2.95 | MFMAs only 1.78 (= the pipe's peak
// for the padded tile grid) | + fragment reads 2.08 | + DMAs 2.56 | DMAs only 1.49 (0.61 us per step: the round trip of one step's
// pieces) | fragment reads only 1.01 | barriers + epilogue 0.24.  With every DMA reading one cached row instead of its real
// address the full kernel takes 2.2 ms and the DMA-only form 0.66 ms: 0.8 ms of the 2.95 is the L2 / fabric side of the 13.6 GB
// the blocks pull per call (10.2 GB of it the 1.2 MB of weights re-streamed by each of the 8640 blocks), not instruction issue;
// moving the DMA issue behind the step's MFMAs changes nothing (2.55 vs 2.56).
// Used for the heatmap heads' first conv (FD:202-212, C -> C) and any other wide stride-1 3x3 conv; stride-2 (pyramid)
// convs and the GEMMs stay on splitmm.hip.
#include <cstdlib>

#include "ff3d_common.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int HC_Y = 4, HC_X = 64, HC_HX = HC_X + 2, HC_HALO = (HC_Y + 2) * HC_HX;   // 396 halo pixels
constexpr int HC_BK = 32, HC_BN = 128, HC_T = 512;
constexpr int HC_ACT = HC_HALO * HC_BK, HC_WT = HC_BN * HC_BK;                      // halves per plane
constexpr int HC_ASLOTS = HC_HALO * 4;                                               // 1584 16-byte slots per plane
constexpr int HC_AIT = (HC_ASLOTS + HC_T - 1) / HC_T;                                // 4 slot rounds
constexpr size_t HC_LDS_BYTES = (size_t)(2 * 2 * HC_ACT + 2 * 2 * HC_WT) * sizeof(_Float16);

struct HaloParams {
  const _Float16 *x_hi, *x_lo, *w_hi, *w_lo;   // x: (B*H*W + 1, C) NHWC + zero row; w: (N + 1, 9, C) + zero row
  const float* bias;
  float* out;                                  // NCHW fp32, or
  _Float16 *out_hi, *out_lo;                   // the (hi, lo') NHWC pair (B*H*W rows of N)
  int B, C, H, W, N, relu;
  unsigned x_zero, w_zero;                     // byte offsets of the zero rows
  Ff3dScale sc;                                // range normalisation (ff3d.h): operand exponents in, output exponent out
  float* out_cl = nullptr;                     // round 5: NHWC fp32 (B*H*W rows of N) - the transposed-tile epilogue writing fp32
                                               // instead of the pair (the camera maps the projection sampler gathers from)
  int w_tiled = 0;                             // round 5 (ff3d_conv3x3_halo_f16x3_tiled): the weight planes arrive K-step-tiled,
                                               // [9 * C / 32][N + 1][32] (tile = tap * C / 32 + c0 / 32; row N = zeros)
  const float* x_f32 = nullptr;                // round 6 (SRC_NCHW instances, ff3d_conv3x3_halo_f16x3_nchwsrc): the activation as the caller's
                                               // NCHW fp32 map; the block converts its halo to (hi, lo') pairs on the way into LDS
  int* x_hint = nullptr;                       // ... its exponent guess {e, max|x| bits, redo, -} + 64 maximum slots (ff3d_split_f16's record);
  int x_redo = 0;                              // sc.a_exp points at x_hint[0]; x_redo = 1: the guarded second launch (exits unless flagged)
};

__device__ __forceinline__ int hc_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }
// Halo rows are read at EVERY alignment (fragment row = halo pixel base + fr, base = any residue mod 16), and the swizzle
// above is conflict-free only for bases that are multiples of 16 (round 3 PMC: SQ_LDS_BANK_CONFLICT 29 % of SQ_LDS_IDX_ACTIVE).
// ds_read_b128 is served in lane groups {fr 0-3, 12-15 of one kq; fr 4-11 of kq ^ 1}: with s(m) the swizzle of row quad m the
// four rows of one bank quarter need {s(m-1), s(m), 1 ^ s(m+1), 1 ^ s(m+2)} distinct for every m - s(m) = 2 (m & 1) is the
// period-4 solution (exhaustive search), whatever the base.
__device__ __forceinline__ int hc_swz_act(int row) { return (row >> 1) & 2; }

// AUX = cache policy bits of the DMA (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX = 0>
__device__ __forceinline__ void hc_glds16(const _Float16* base, unsigned byte_off, _Float16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(base) + byte_off,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, AUX);
}

// Tile geometry of a block (256 pixels x 128 channels either way; a wave = 64 pixels = 4 M-tiles of 16 consecutive pixels of a row):
//   GEO 0: 4 rows x 64 pixels - wave wr = row wr, M-tile i at x = 16 i.                 Halo 6 x 66 = 396 pixels.
//   GEO 1: 8 rows x 32 pixels - wave wr = rows 2 wr, 2 wr + 1, M-tile i = row (i >> 1), x = 16 (i & 1).  Halo 10 x 34 = 340.
// The launcher picks the geometry that pads the map less (468 x 468: 8 x 64 = 512 columns against 15 x 32 = 480; 180 x 180: GEO 0).
template <int GEO>
struct HcGeo {
  static constexpr int TY = GEO ? 8 : 4, TX = GEO ? 32 : 64, HX = TX + 2, HALO = (TY + 2) * HX;
  static constexpr int ACT = HALO * HC_BK, ASLOTS = HALO * 4, AIT = (ASLOTS + HC_T - 1) / HC_T;
  static constexpr size_t LDS_BYTES = (size_t)(2 * 2 * ACT + 2 * 2 * HC_WT) * sizeof(_Float16);
  __device__ static __forceinline__ int row(int wr, int i) { return GEO ? 2 * wr + (i >> 1) : wr; }
  __device__ static __forceinline__ int col(int i) { return GEO ? (i & 1) * 16 : i * 16; }
};

// Epilogue: D row = 4*kq + r (pixel x offset inside the M-tile), col = fr (output channel); TR: transposed.
template <bool TR, int GEO = 0>
__device__ __forceinline__ void hc_epilogue(const HaloParams& p, f32x4 (&acc_m)[4][4], f32x4 (&acc_x)[4][4], unsigned lid, int tid,
                                            int b, int ty0, int tx0, int n0, int wr, int wc, int fr, int kq, int lane) {
  const int e_a = ff3d_ld_exp(p.sc.a_exp);
  const float sc_in = ff3d_pow2(e_a + ff3d_ld_exp(p.sc.w_exp));
  float sc_out = 1.f;
  if (p.sc.out_exp) {
    const int e_out = ff3d_out_exp(p.sc, e_a, false, INFINITY);
    if (!p.out) sc_out = ff3d_pow2(-e_out);
    if (lid == 0 && tid == 0) *p.sc.out_exp = e_out;
  }
  using G = HcGeo<GEO>;
  if (TR) {   // pair output: lane = pixel x (column fr of the transposed tile), channels n .. n + 3
    const bool n4 = (p.N & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = ty0 + G::row(wr, i), x = tx0 + G::col(i) + fr;
      if (x >= p.W || y >= p.H) continue;
      const long long pix = ((long long)b * p.H + y) * p.W + x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + kq * 4;
        if (n >= p.N) continue;
        _Float16 h[4], l[4];
        float vf[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc_m[i][j][r] + acc_x[i][j][r] * (1.f / 2048.f), sc_in, (p.bias && n + r < p.N) ? p.bias[n + r] : 0.f);
          if (p.relu) v = fmaxf(v, 0.f);
          vf[r] = v;
          v *= sc_out;
          h[r] = (_Float16)v;
          l[r] = (_Float16)((v - (float)h[r]) * 2048.f);
        }
        const long long o = pix * p.N + n;
        if (p.out_cl) {                   // channels-last fp32: 4 consecutive channels of one pixel = one 16-byte store
          if (n4) {
            *reinterpret_cast<float4*>(p.out_cl + o) = make_float4(vf[0], vf[1], vf[2], vf[3]);
          } else {
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) p.out_cl[o + r] = vf[r];
          }
          continue;
        }
        if (n4) {
          *reinterpret_cast<uint2*>(p.out_hi + o) = *reinterpret_cast<uint2*>(h);
          *reinterpret_cast<uint2*>(p.out_lo + o) = *reinterpret_cast<uint2*>(l);
        } else {
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) p.out_hi[o + r] = h[r], p.out_lo[o + r] = l[r];
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wc * 64 + j * 16 + fr;
    if (n >= p.N) continue;
    const float bj = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = ty0 + G::row(wr, i), x = tx0 + G::col(i) + kq * 4;
      if (y >= p.H) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = fmaf(acc_m[i][j][r] + acc_x[i][j][r] * (1.f / 2048.f), sc_in, bj);
        if (p.relu) v[r] = fmaxf(v[r], 0.f);
      }
      if (p.out) {
        float* o = p.out + (((long long)b * p.N + n) * p.H + y) * p.W + x;
        if (x + 3 < p.W && (p.W & 3) == 0) {
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (x + r < p.W) o[r] = v[r];
        }
      } else {
        // (hi, lo') NHWC pair: lanes 2k / 2k+1 (neighbouring channels, same 4 pixels) swap two pixels each so that every
        // lane stores two 4-byte channel pairs
        const bool odd = lane & 1;
        unsigned hs[4], ls[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float vs = v[r] * sc_out;
          const _Float16 h = (_Float16)vs;
          const _Float16 l = (_Float16)((vs - (float)h) * 2048.f);
          hs[r] = __builtin_bit_cast(unsigned short, h);
          ls[r] = __builtin_bit_cast(unsigned short, l);
        }
        const unsigned send_h = odd ? (hs[0] | (hs[1] << 16)) : (hs[2] | (hs[3] << 16));
        const unsigned send_l = odd ? (ls[0] | (ls[1] << 16)) : (ls[2] | (ls[3] << 16));
        const unsigned recv_h = __shfl_xor(send_h, 1), recv_l = __shfl_xor(send_l, 1);
        const int r0 = odd ? 2 : 0, nc = n & ~1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const unsigned mine_h = hs[r0 + q], mine_l = ls[r0 + q];
          const unsigned other_h = (recv_h >> (16 * q)) & 0xffffu, other_l = (recv_l >> (16 * q)) & 0xffffu;
          const unsigned ph = odd ? (other_h | (mine_h << 16)) : (mine_h | (other_h << 16));
          const unsigned pl = odd ? (other_l | (mine_l << 16)) : (mine_l | (other_l << 16));
          if (x + r0 + q < p.W) {
            const long long o = (((long long)b * p.H + y) * p.W + x + r0 + q) * p.N + nc;
            *reinterpret_cast<unsigned*>(p.out_hi + o) = ph;
            *reinterpret_cast<unsigned*>(p.out_lo + o) = pl;
          }
        }
      }
    }
  }
}

// TR (pair output): transposed accumulators (operands of the MFMA swapped), so a lane holds 4 consecutive CHANNELS of one
// pixel - the NHWC planes then take one 8-byte store per plane and tile instead of two 4-byte stores after a lane exchange.
// ABL: timing ablations behind the numbers above (tuning only, WRONG results): 1 no MFMA, 2 no DMA, 4 no fragment reads,
// 8 every DMA reads one cached row; 16 (correct results) the rounds 1-2 halo swizzle hc_swz instead of hc_swz_act.
template <bool TR, int ABL, bool PASS_MAJOR, int GEO, bool SRC_NCHW = false, int NVAR = 0>
__device__ __forceinline__ void hc_body(const HaloParams& p, unsigned bid, unsigned nblk) {
  using G = HcGeo<GEO>;
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  _Float16* const s_act = lds;                           // [2 buffers][2 planes][G::ACT]
  _Float16* const s_wt = lds + 2 * 2 * G::ACT;           // [2 buffers][2 planes][HC_WT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_x = (p.W + G::TX - 1) / G::TX, tiles_y = (p.H + G::TY - 1) / G::TY, n_tiles = (p.N + HC_BN - 1) / HC_BN;
  const unsigned lid = ff3d_xcd_remap(bid, nblk);
  int nt = (int)(lid % n_tiles), sp = (int)(lid / n_tiles);
  if ((ABL & 128) && nblk % (FF3D_NUM_XCD * n_tiles) == 0) {
    // ABL & 128 (correct results): N-tile-major order inside an XCD's chunk - the 32 CUs of an XCD stream ONE N-tile's weights at a
    // time (1.2 MB of the 4 MB L2 instead of 2.4 MB); each halo is then fetched twice at distant times instead of twice at once
    const unsigned per_xcd = nblk / FF3D_NUM_XCD, per_nt = per_xcd / n_tiles, xcd = bid % FF3D_NUM_XCD, i = bid / FF3D_NUM_XCD;
    nt = (int)(i / per_nt);
    sp = (int)(xcd * per_nt + i % per_nt);
  }
  const int b = sp / (tiles_x * tiles_y), t = sp % (tiles_x * tiles_y);
  const int ty0 = (t / tiles_x) * G::TY, tx0 = (t % tiles_x) * G::TX, n0 = nt * HC_BN;

  // ---- DMA slot geometry (chunk / tap invariant)
  unsigned a_off[G::AIT];
#pragma unroll
  for (int it = 0; it < G::AIT; ++it) {
    const int s = it * HC_T + tid, px = min(s >> 2, G::HALO - 1), ly = px / G::HX, lx = px - ly * G::HX;
    const int gy = ty0 + ly - 1, gx = tx0 + lx - 1;
    const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    a_off[it] = (in ? (unsigned)(((b * p.H + gy) * p.W + gx) * p.C) * 2u : p.x_zero) + (unsigned)(((s & 3) ^ ((ABL & 16) ? hc_swz(px) : hc_swz_act(px))) * 16);
  }
  unsigned w_off;
  {
    const int row = tid >> 2, n = n0 + row;              // 512 slots = 128 rows x 4 chunks: one per thread
    w_off = (n < p.N ? (unsigned)(n * 9 * p.C) * 2u : p.w_zero) + (unsigned)(((tid & 3) ^ hc_swz(row)) * 16);
    if (p.w_tiled) w_off = (unsigned)min(n, p.N) * 64u + (unsigned)(((tid & 3) ^ hc_swz(row)) * 16);
  }
  const unsigned w_tile_bytes = (unsigned)(p.N + 1) * 64u;
  // ---- SRC_NCHW: slot s = (halo pixel s >> 2, 16-byte piece s & 3) as for the DMA; the piece holds channels 8 * ((s & 3) ^ swizzle) .. + 7
  //      of the chunk.  n_off = byte offset of (that channel group, the pixel) inside the frame's (C, H, W) fp32 block, ~0u for padding.
  unsigned n_off[G::AIT];
  const int HWp = p.H * p.W;
  const char* const xb = reinterpret_cast<const char*>(p.x_f32) + (SRC_NCHW ? (size_t)b * p.C * HWp * 4 : 0);
  float nsc = 1.f, n_amax = 0.f;
  if (SRC_NCHW) {
    if (p.x_redo && p.x_hint && p.x_hint[2] == 0) return;       // guarded second launch: the guessed exponent held (uniform: before any barrier)
    nsc = ff3d_pow2(-ff3d_ld_exp(p.sc.a_exp));
#pragma unroll
    for (int it = 0; it < G::AIT; ++it) {
      const int s = it * HC_T + tid;
      // NVAR & 8: pixel-fastest slots - a wave's request is 64 consecutive halo pixels of ONE channel (whole 128-byte lines) and its
      // LDS writes are 64 bytes apart; default: the DMA's slot order (16 pixels x 4 channel groups per wave, linear LDS writes)
      const int px = (NVAR & 8) ? min(s, G::ASLOTS - 1) % G::HALO : min(s >> 2, G::HALO - 1);
      const int grp = (NVAR & 8) ? min(s, G::ASLOTS - 1) / G::HALO : ((s & 3) ^ hc_swz_act(px));
      const int ly = px / G::HX, lx = px - ly * G::HX;
      const int gy = ty0 + ly - 1, gx = tx0 + lx - 1;
      const bool in = s < G::ASLOTS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      n_off[it] = in ? (unsigned)((grp * 8) * HWp + gy * p.W + gx) * 4u : ~0u;
    }
  }
  // A slot round is handled as two half rounds of 4 channels (8 bytes per plane): the values of half round u are requested right after
  // the first MFMA pass of tap u (landed at tap u + 1's vmcnt(0)) and converted + written to LDS after the first MFMA pass of tap u + 1 -
  // 4 registers in flight, the conversion's ~30 VALU instructions in the shadow of that step's MFMAs.
  float nl[4], nl2[4];                                    // (nl2: NVAR & 32, two request steps in flight)
  // buffer loads: the frame's (C, H, W) block as a raw buffer (base in SGPRs, the channel offset as the scalar offset, n_off as the
  // per-lane offset) - no 64-bit address arithmetic in VGPRs, and the padding slots (n_off = ~0u >= num_records) read as 0 by the
  // buffer's range check: no branch.
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(SRC_NCHW ? xb : nullptr), 0, 0x80000000, 0x00020000);
  auto nchw_load_to = [&](int u, int c0, float (&r)[4]) {
    const int it = u >> 1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      r[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, n_off[it], (unsigned)(c0 + 4 * (u & 1) + k) * (unsigned)HWp * 4u,
                                                                          (NVAR & 64) ? 2 : 0));      // NVAR & 64: non-temporal requests
  };
  auto nchw_load = [&](int u, int c0) { nchw_load_to(u, c0, nl); };
  auto nchw_store_from = [&](int u, int buf, const float (&nl)[4]) {
    const int it = u >> 1;
    if (it * HC_T + tid >= G::ASLOTS) return;
    using half4 = __attribute__((ext_vector_type(4))) _Float16;
    half4 h, l;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v = nl[k] * nsc;
      const _Float16 hk = (_Float16)v;
      h[k] = hk;
      l[k] = (_Float16)((v - (float)hk) * 2048.f);
    }
    n_amax = fmaxf(fmaxf(n_amax, fmaxf(fabsf(nl[0]), fabsf(nl[1]))), fmaxf(fabsf(nl[2]), fabsf(nl[3])));      // max|x| of the pass (the guess's check)
    _Float16* dst = s_act + buf * 2 * G::ACT + (it * HC_T + tid) * 8 + 4 * (u & 1);
    if (NVAR & 8) {
      const int sl = it * HC_T + tid, px = sl % G::HALO, grp = sl / G::HALO;
      dst = s_act + buf * 2 * G::ACT + px * HC_BK + ((grp ^ hc_swz_act(px)) * 8) + 4 * (u & 1);
    }
    *reinterpret_cast<half4*>(dst) = h;
    *reinterpret_cast<half4*>(dst + G::ACT) = l;
  };
  auto nchw_store = [&](int u, int buf) { nchw_store_from(u, buf, nl); };
  auto dma_act = [&](int it, int c0, int buf) {          // one slot round of the halo of channel chunk c0
    if (ABL & 2) return;
    if (it * HC_T + tid < G::ASLOTS) {
      _Float16* dst = s_act + buf * 2 * G::ACT + (it * HC_T + wave * 64) * 8;   // wave-uniform; the DMA adds lane * 16 B
      const unsigned o = (ABL & 8) ? p.x_zero + (unsigned)(lane & 3) * 16u : a_off[it] + (unsigned)c0 * 2u;
      hc_glds16<(ABL & 32) ? 2 : 0>(p.x_hi, o, dst);     // ABL & 32 (correct results): the halo stream non-temporal (each piece is read by
      hc_glds16<(ABL & 32) ? 2 : 0>(p.x_lo, o, dst + G::ACT);   // two blocks, the weights by all 8 640: keep those in L2)
    }
  };
  auto dma_wt = [&](int tap, int c0, int buf) {
    if (ABL & 2) return;
    _Float16* dst = s_wt + buf * 2 * HC_WT + (wave * 64) * 8;
    const unsigned o = (ABL & 8)    ? p.w_zero + (unsigned)(lane & 3) * 16u
                       : p.w_tiled ? w_off + (unsigned)((tap * p.C + c0) >> 5) * w_tile_bytes
                                   : w_off + (unsigned)(tap * p.C + c0) * 2u;
    hc_glds16(p.w_hi, o, dst);
    hc_glds16(p.w_lo, o, dst + HC_WT);
  };

  f32x4 acc_m[4][4], acc_x[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc_m[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}, acc_x[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int b_rd[4];                                           // weight fragment offsets (tap invariant)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rb = wc * 64 + j * 16 + fr;
    b_rd[j] = rb * HC_BK + ((kq ^ hc_swz(rb)) * 8);
  }

  const int nchunks = p.C / HC_BK;
  if (SRC_NCHW) {
#pragma unroll
    for (int u = 0; u < 2 * G::AIT; ++u) {
      nchw_load(u, 0);
      nchw_store(u, 0);
    }
  } else {
#pragma unroll
    for (int it = 0; it < G::AIT; ++it) dma_act(it, 0, 0);
  }
  dma_wt(0, 0, 0);
  // ABL & 64 (correct results): static priority for the second-dispatched half of the block (MI355X_MICROARCH.md, "two waves
  // per SIMD", item 4: waves 4-7 are the arbitration losers of every segment)
  if ((ABL & 64) && wave >= 4) __builtin_amdgcn_s_setprio(1);
  int wbuf = 0;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * HC_BK;
    const _Float16* act = s_act + (ch & 1) * 2 * G::ACT;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (SRC_NCHW && (NVAR & 32) && tap >= 1 && tap - 1 < 2 * G::AIT)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // everything but the previous step's 4 activation requests
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();           // weights of this step (and, at tap 0, the chunk's halo) landed; previous reads retired
      if (SRC_NCHW && (NVAR & 128) && (NVAR & 1) && ch + 1 < nchunks) {      // NVAR & 128: the activation requests ahead of the weight DMAs
        if (tap >= 1 && tap - 1 < 2 * G::AIT) nchw_store(tap - 1, (ch + 1) & 1);
        if (tap < 2 * G::AIT) nchw_load(tap, c0 + HC_BK);
      }
      if (tap < 8)
        dma_wt(tap + 1, c0, wbuf ^ 1);
      else if (ch + 1 < nchunks)
        dma_wt(0, c0 + HC_BK, wbuf ^ 1);
      if (SRC_NCHW && (NVAR & 32)) {
        // two request steps in flight: half round `tap` requested here (into the register set tap & 1, behind the weight DMAs), half round
        // tap - 1 converted behind the first MFMA pass below - a request has one step + one MFMA pass to come back
        if (tap < 2 * G::AIT) {
          if (ch + 1 < nchunks) {
            if (tap & 1) nchw_load_to(tap, c0 + HC_BK, nl2); else nchw_load_to(tap, c0 + HC_BK, nl);
          } else {                                      // (keeps the vmcnt bookkeeping static: 4 requests that read nothing)
            const unsigned keep = n_off[0];
            n_off[0] = ~0u;
            if (tap & 1) nchw_load_to(0, 0, nl2); else nchw_load_to(0, 0, nl);
            n_off[0] = keep;
          }
        }
      } else if (SRC_NCHW) {
        // (the next halo's half rounds are requested / converted behind the first MFMA pass below; NVAR & 1: here, at the top of the step)
        if ((NVAR & 1) && !(NVAR & 128) && ch + 1 < nchunks) {
          if (!(NVAR & 4) && tap >= 1 && tap - 1 < 2 * G::AIT) nchw_store(tap - 1, (ch + 1) & 1);
          if (!(NVAR & 2) && tap < 2 * G::AIT) nchw_load(tap, c0 + HC_BK);
        }
      } else if (tap < G::AIT && ch + 1 < nchunks) {
        dma_act(tap, c0 + HC_BK, (ch + 1) & 1);   // next halo, one slot round per tap
      }
      const int dy = tap / 3, dx = tap - dy * 3;
      const _Float16* wt = s_wt + wbuf * 2 * HC_WT;
      half8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (ABL & 4) {
          ah[i] = al[i] = bh[i] = bl[i] = half8{1, 1, 1, 1, 1, 1, 1, 1};
          continue;
        }
        const int hp = (G::row(wr, i) + dy) * G::HX + G::col(i) + dx + fr;   // halo pixel of this lane for (tap, M-tile i)
        const int ao = hp * HC_BK + ((kq ^ ((ABL & 16) ? hc_swz(hp) : hc_swz_act(hp))) * 8);
        ah[i] = *reinterpret_cast<const half8*>(act + ao);
        al[i] = *reinterpret_cast<const half8*>(act + G::ACT + ao);
        if (SRC_NCHW) {           // hc_swz(rb) depends on fr only (rb = wc * 64 + i * 16 + fr): one register + immediate offsets
          bh[i] = *reinterpret_cast<const half8*>(wt + b_rd[0] + i * 16 * HC_BK);
          bl[i] = *reinterpret_cast<const half8*>(wt + HC_WT + b_rd[0] + i * 16 * HC_BK);
        } else {
          bh[i] = *reinterpret_cast<const half8*>(wt + b_rd[i]);
          bl[i] = *reinterpret_cast<const half8*>(wt + HC_WT + b_rd[i]);
        }
      }
      if (ABL & 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]), "v"(bh[i]), "v"(bl[i]));
      } else {
        // PASS-MAJOR order (round 3): the two cross-term MFMAs of a tile depend on each other through acc_x, and a
        // v_mfma_f32_16x16x32_f16 issued right behind the one that produces its C operand waits out the full 8-pass latency
        // instead of the 16-cycle issue interval.  Tile-major order (m, x, x per tile - rounds 1-2, and what the "56 %
        // MFMA-busy whatever the tile shape" of splitmm.hip's header came from) leaves one such stall per tile; here the 16
        // tiles take the hi*hi pass, then the first cross pass, then the second: dependent instructions are 16 MFMAs apart.
        // FF3D_MFMA_ORDER=tile restores the old order (A/B runs).
        if (PASS_MAJOR) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc_m[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc_m[i][j], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc_m[i][j], 0, 0, 0);
          if (SRC_NCHW && (NVAR & 32) && ch + 1 < nchunks && tap >= 1 && tap - 1 < 2 * G::AIT) {
            if ((tap - 1) & 1) nchw_store_from(tap - 1, (ch + 1) & 1, nl2); else nchw_store_from(tap - 1, (ch + 1) & 1, nl);
          }
          if (SRC_NCHW && !(NVAR & 1) && !(NVAR & 32) && ch + 1 < nchunks) {
            if (!(NVAR & 4) && tap >= 1 && tap - 1 < 2 * G::AIT) nchw_store(tap - 1, (ch + 1) & 1);
            if (!(NVAR & 2) && tap < 2 * G::AIT) nchw_load(tap, c0 + HC_BK);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc_x[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc_x[i][j], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc_x[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc_x[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc_x[i][j], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc_x[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (TR) {
                acc_m[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc_m[i][j], 0, 0, 0);
                acc_x[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc_x[i][j], 0, 0, 0);
                acc_x[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc_x[i][j], 0, 0, 0);
              } else {
                acc_m[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc_m[i][j], 0, 0, 0);
                acc_x[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc_x[i][j], 0, 0, 0);
                acc_x[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc_x[i][j], 0, 0, 0);
              }
            }
        }
      }
      wbuf ^= 1;
    }
  }

  if (SRC_NCHW && p.x_hint && !p.x_redo) {
    // every element of the map is some block's tile interior, so the 64 slots end up holding max|x|: one fire-and-forget atomicMax per wave
    // (|x| bit patterns order like unsigned ints), as in the conversion pass this launch replaces (splitmm.hip: wave_amax)
    float m = n_amax;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0 && m > 0.f)
      __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(p.x_hint) + (1 + ((lid * 8u + (unsigned)wave) & 63u)) * 64, __float_as_uint(m),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  hc_epilogue<TR, GEO>(p, acc_m, acc_x, lid, tid, b, ty0, tx0, n0, wr, wc, fr, kq, lane);
}

template <bool TR, int ABL = 0, bool PASS_MAJOR = true, int GEO = 0>
__global__ __launch_bounds__(HC_T, 1) void conv3x3_halo_f16x3_kernel(HaloParams p) {
  hc_body<TR, ABL, PASS_MAJOR, GEO>(p, blockIdx.x, gridDim.x);
}

// Round 6: the same convolution reading the caller's NCHW fp32 map (HaloParams::x_f32) - the NCHW -> NHWC-pair conversion pass (read 4 +
// write 4 bytes per element, 0.43 - 0.46 ms per 32 x 256 x 180 x 180 map) folded into the halo staging: each thread requests 4 channels of
// one halo pixel per tap (buffer_load_dword, the frame as a raw buffer: padding = out-of-range offsets that read as 0), converts them to
// (hi, lo') at the next tap and writes 8 + 8 bytes into the halo buffer the DMA would have filled.  Same bits as conversion pass + conv.
// Measured (tools/experiments/exp_halo_nchw.py, profiles/r06_nc_halo_nchw_source_ab.txt): 3.16 - 3.22 ms against 3.28 - 3.47 ms for
// conversion + conv back to back; the requests cost 0.5 ms (2.67 ms without them) against the DMA's 0.3 - and no more flight time helps
// (two register sets in flight: level), while every form that keeps more in flight (LDS-staged fp32 by 4- or 16-byte DMA) spills: the loop
// has 229 of 256 registers taken, and a spilled register comes back through vmcnt, which ends the flight.
// The exponent: the conversion pass's guess-verify-redo record (ff3d.h) moves to the conv - it runs with the guessed exponent while its
// requests measure max|x|; hc_nchw_verify_kernel checks the guess; a second launch of the conv recomputes only when flagged.
// NVAR (tuning): bit 0 = request / convert at the top of the step instead of behind the first MFMA pass; bit 1 (WRONG results) = no
// requests inside the loop; bit 2 (WRONG results) = no conversion + LDS write inside the loop.
template <bool TR, int GEO = 0, int NVAR = 0>
__global__ __launch_bounds__(HC_T, 1) void conv3x3_halo_nchw_f16x3_kernel(HaloParams p) {
  hc_body<TR, 0, true, GEO, true, NVAR>(p, blockIdx.x, gridDim.x);
}

// The guarded second launch: 512 blocks that leave at once when the guess held (a full grid of 8 640 leaving blocks costs ~0.1 ms), and walk
// the tiles otherwise.
template <bool TR, int GEO = 0, int NVAR = 0>
__global__ __launch_bounds__(HC_T, 1) void conv3x3_halo_nchw_redo_f16x3_kernel(HaloParams p, unsigned nblk) {
  if (p.x_hint[2] == 0) return;
  for (unsigned bid = blockIdx.x; bid < nblk; bid += gridDim.x) {
    hc_body<TR, 0, true, GEO, true, NVAR>(p, bid, nblk);
    __syncthreads();                       // the next tile's prologue rewrites the LDS buffers
  }
}

// One wave: the check of splitmm.hip's split_verify_kernel on the same record - accept the guessed exponent iff 2^5 <= max|x| * 2^-e < 2^15,
// otherwise take the exponent that puts max|x| into [2^13, 2^14) and flag the redo launch.  Resets the maximum slots.
__global__ __launch_bounds__(64) void hc_nchw_verify_kernel(int* __restrict__ hint) {
  unsigned* slot = reinterpret_cast<unsigned*>(hint) + (1 + threadIdx.x) * 64;
  float amax = __uint_as_float(*slot);
  *slot = 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  if (threadIdx.x != 0) return;
  const int e_guess = hint[0];
  int e_final = e_guess, redo = 0;
  if (amax > 0.f) {
    const int e_star = ff3d_bound_exp(amax), d = e_guess - e_star;
    if (d < -1 || d > 8) e_final = e_star, redo = 1;
  }
  hint[0] = e_final, hint[1] = (int)__float_as_uint(amax), hint[2] = redo;
}

// Round 6 (the 8 x 32 geometry's default; FF3D_HALO_TAP2=0 restores the form above): TWO filter taps per barrier.  The kernel above synchronises once per
// (chunk, tap) step - 72 x per block: s_waitcnt vmcnt(0) + s_barrier, ~0.16 us each by the ablations in the header ("barriers +
// epilogue 0.24" + part of the DMA wait) - because the weights of ONE tap are what the double buffer holds.  Here a buffer holds the
// weights of two taps (2 x 2 x 32 KiB; with the 8 x 32 halo 149 KiB - the 4 x 64 halo does not leave the room), a chunk is the five
// steps (0 1)(2 3)(4 5)(6 7)(8): 40 barriers per block, the same DMA pieces, fragment reads and MFMAs.  Same-box A/B
// (profiles/r06_h2_halo_tap2_ab.txt): 232 x 400 x 48 maps (configs[2]'s camera conv) 11.95 - 12.06 vs 12.96 - 13.03 ms, 468 x 468 x 8
// (configs[4]) 4.78 vs 5.05 - 5.07 ms, results bit-identical in error (5.0e-7 / 6.3e-7); at 180 x 180 the 8 x 32 geometry with it
// (2.95 - 2.98 ms) now equals the 4 x 64 form per padded pixel but pads 2.2 % more (2.89 ms): the 180 x 180 maps stay on 4 x 64.
// Tried for the 180 x 180 maps: a 4 x 62-stored-pixel geometry (tile grid advancing by 62, halo 6 x 64 = 96 KiB + the 64 KiB of two
// taps' weights = exactly the 160 KiB of a CU, the same three tile columns) with two taps per barrier - 3.02 - 3.05 vs 3.00 - 3.03 ms
// for the pair output, default step 1301.2 - 1301.8 vs 1294.5 - 1298.6 frames/s: level.  The 4 x 64 form is not bound by its barrier
// count; what two taps per barrier removed in the 8 x 32 geometry was that geometry's own 7 % deficit per pixel.  Not kept.
template <bool TR, int GEO>
__device__ __forceinline__ void hc_body_tap2(const HaloParams& p, unsigned bid, unsigned nblk) {
  using G = HcGeo<GEO>;
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  _Float16* const s_act = lds;                           // [2 buffers][2 planes][G::ACT]
  _Float16* const s_wt = lds + 2 * 2 * G::ACT;           // [2 buffers][2 taps][2 planes][HC_WT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_x = (p.W + G::TX - 1) / G::TX, tiles_y = (p.H + G::TY - 1) / G::TY, n_tiles = (p.N + HC_BN - 1) / HC_BN;
  const unsigned lid = ff3d_xcd_remap(bid, nblk);
  const int nt = (int)(lid % n_tiles), sp_ = (int)(lid / n_tiles);
  const int b = sp_ / (tiles_x * tiles_y), t = sp_ % (tiles_x * tiles_y);
  const int ty0 = (t / tiles_x) * G::TY, tx0 = (t % tiles_x) * G::TX, n0 = nt * HC_BN;

  unsigned a_off[G::AIT];
#pragma unroll
  for (int it = 0; it < G::AIT; ++it) {
    const int s = it * HC_T + tid, px = min(s >> 2, G::HALO - 1), ly = px / G::HX, lx = px - ly * G::HX;
    const int gy = ty0 + ly - 1, gx = tx0 + lx - 1;
    const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    a_off[it] = (in ? (unsigned)(((b * p.H + gy) * p.W + gx) * p.C) * 2u : p.x_zero) + (unsigned)(((s & 3) ^ hc_swz_act(px)) * 16);
  }
  unsigned w_off;
  {
    const int row = tid >> 2, n = n0 + row;
    w_off = (n < p.N ? (unsigned)(n * 9 * p.C) * 2u : p.w_zero) + (unsigned)(((tid & 3) ^ hc_swz(row)) * 16);
    if (p.w_tiled) w_off = (unsigned)min(n, p.N) * 64u + (unsigned)(((tid & 3) ^ hc_swz(row)) * 16);
  }
  const unsigned w_tile_bytes = (unsigned)(p.N + 1) * 64u;
  auto dma_act = [&](int it, int c0, int buf) {
    if (it * HC_T + tid < G::ASLOTS) {
      _Float16* dst = s_act + buf * 2 * G::ACT + (it * HC_T + wave * 64) * 8;
      const unsigned o = a_off[it] + (unsigned)c0 * 2u;
      hc_glds16(p.x_hi, o, dst);
      hc_glds16(p.x_lo, o, dst + G::ACT);
    }
  };
  auto dma_wt = [&](int tap, int c0, int buf, int slot) {
    _Float16* dst = s_wt + (buf * 2 + slot) * 2 * HC_WT + (wave * 64) * 8;
    const unsigned o = p.w_tiled ? w_off + (unsigned)((tap * p.C + c0) >> 5) * w_tile_bytes : w_off + (unsigned)(tap * p.C + c0) * 2u;
    hc_glds16(p.w_hi, o, dst);
    hc_glds16(p.w_lo, o, dst + HC_WT);
  };

  f32x4 acc_m[4][4], acc_x[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc_m[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}, acc_x[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int b_rd[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rb = wc * 64 + j * 16 + fr;
    b_rd[j] = rb * HC_BK + ((kq ^ hc_swz(rb)) * 8);
  }

  const int nchunks = p.C / HC_BK;
#pragma unroll
  for (int it = 0; it < G::AIT; ++it) dma_act(it, 0, 0);
  dma_wt(0, 0, 0, 0);
  dma_wt(1, 0, 0, 1);
  int wbuf = 0;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * HC_BK;
    const _Float16* act = s_act + (ch & 1) * 2 * G::ACT;
#pragma unroll
    for (int sp = 0; sp < 5; ++sp) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();           // this step's two taps (and, at step 0, the chunk's halo) landed; the previous step's reads retired
      if (sp < 4) {
        dma_wt(2 * sp + 2, c0, wbuf ^ 1, 0);
        if (2 * sp + 3 < 9) dma_wt(2 * sp + 3, c0, wbuf ^ 1, 1);
      } else if (ch + 1 < nchunks) {
        dma_wt(0, c0 + HC_BK, wbuf ^ 1, 0);
        dma_wt(1, c0 + HC_BK, wbuf ^ 1, 1);
      }
      if (sp < G::AIT && ch + 1 < nchunks) dma_act(sp, c0 + HC_BK, (ch + 1) & 1);
#pragma unroll
      for (int slot = 0; slot < 2; ++slot) {
        const int tap = 2 * sp + slot;
        if (tap >= 9) break;
        const int dy = tap / 3, dx = tap - dy * 3;
        const _Float16* wt = s_wt + (wbuf * 2 + slot) * 2 * HC_WT;
        half8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int hp = (G::row(wr, i) + dy) * G::HX + G::col(i) + dx + fr;
          const int ao = hp * HC_BK + ((kq ^ hc_swz_act(hp)) * 8);
          ah[i] = *reinterpret_cast<const half8*>(act + ao);
          al[i] = *reinterpret_cast<const half8*>(act + G::ACT + ao);
          bh[i] = *reinterpret_cast<const half8*>(wt + b_rd[i]);
          bl[i] = *reinterpret_cast<const half8*>(wt + HC_WT + b_rd[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc_m[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc_m[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc_m[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc_x[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc_x[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc_x[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc_x[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc_x[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc_x[i][j], 0, 0, 0);
      }
      wbuf ^= 1;
    }
  }
  hc_epilogue<TR, GEO>(p, acc_m, acc_x, lid, tid, b, ty0, tx0, n0, wr, wc, fr, kq, lane);
}

template <bool TR, int GEO>
__global__ __launch_bounds__(HC_T, 1) void conv3x3_halo_tap2_f16x3_kernel(HaloParams p) {
  hc_body_tap2<TR, GEO>(p, blockIdx.x, gridDim.x);
}
template <int GEO>
constexpr size_t hc_tap2_lds_bytes() {
  return (size_t)(2 * 2 * HcGeo<GEO>::ACT + 2 * 2 * 2 * HC_WT) * sizeof(_Float16);
}

// Several convolutions of ONE shape (own inputs, weights, outputs, exponents) in one launch: the three heatmap heads of the
// multi-stage head (FD:587-668 computes them up front).  Round 3: grid arithmetic - at 4 frames a conv is 1 080 blocks on 256
// CUs = 4.2 rounds (5 with the last 22 % full); three of them in one grid are 12.7 rounds (13).
constexpr int HC_MAX_GROUP = 4;
struct HaloGroup {
  HaloParams p[HC_MAX_GROUP];
  unsigned per;                                          // blocks per member
};
template <bool TR, int GEO>
__global__ __launch_bounds__(HC_T, 1) void conv3x3_halo_group_kernel(HaloGroup gp) {
  const unsigned g = blockIdx.x / gp.per;
  hc_body<TR, 0, true, GEO>(gp.p[g], blockIdx.x - g * gp.per, gp.per);
}
template <bool TR, int GEO>
__global__ __launch_bounds__(HC_T, 1) void conv3x3_halo_group_tap2_kernel(HaloGroup gp) {
  const unsigned g = blockIdx.x / gp.per;
  hc_body_tap2<TR, GEO>(gp.p[g], blockIdx.x - g * gp.per, gp.per);
}


#ifdef FF3D_BUILD_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------------------
// Round 4 experiment (FF3D_HALO_WREG=1): the weights never pass through the LDS.  Hypothesis: the kernel above is bound by its
// FEED - 13.6 GB per launch through the LDS DMA (10.2 GB of it weights), one s_waitcnt vmcnt(0) + barrier per filter tap - not by
// MFMA scheduling or LDS banks (rounds 2-3).  Here only the halo (3.4 GB per launch) is DMAed; every wave loads the weight
// fragments of its own 32 output channels straight from L1 / L2 into registers, one tap ahead (plain 16-byte global loads:
// a different path - vector L1 - than the LDS DMA and the fragment reads), and the block synchronises once per 32-channel chunk
// instead of once per tap.
//   block = 512 threads = 8 waves = 2 (row pairs) x 4 (32-channel quarters); tile 4 x 64 pixels x 128 channels as above
//   wave tile = 2 rows x 64 pixels x 32 channels = 8 x 2 MFMA tiles x 2 accumulators (128 registers)
//   per (chunk, tap) step and wave: 4 global loads (w_hi, w_lo' of two 16-channel tiles; the two row-pair waves of a quarter share
//   them through L1), 16 LDS fragment reads (as many per MFMA as above), 48 MFMAs; LDS = the double-buffered halo only (99 KiB)
template <bool TR>
__global__ __launch_bounds__(HC_T, 1) void conv3x3_halo_wreg_f16x3_kernel(HaloParams p) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  _Float16* const s_act = lds;                           // [2 buffers][2 planes][HC_ACT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int wr2 = wave >> 2, wq = wave & 3;
  const int tiles_x = (p.W + HC_X - 1) / HC_X, tiles_y = (p.H + HC_Y - 1) / HC_Y, n_tiles = (p.N + HC_BN - 1) / HC_BN;
  const unsigned lid = ff3d_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = (int)(lid % n_tiles);
  const int sp = (int)(lid / n_tiles), b = sp / (tiles_x * tiles_y), t = sp % (tiles_x * tiles_y);
  const int ty0 = (t / tiles_x) * HC_Y, tx0 = (t % tiles_x) * HC_X, n0 = nt * HC_BN;

  unsigned a_off[HC_AIT];
#pragma unroll
  for (int it = 0; it < HC_AIT; ++it) {
    const int s = it * HC_T + tid, px = min(s >> 2, HC_HALO - 1), ly = px / HC_HX, lx = px - ly * HC_HX;
    const int gy = ty0 + ly - 1, gx = tx0 + lx - 1;
    const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    a_off[it] = (in ? (unsigned)(((b * p.H + gy) * p.W + gx) * p.C) * 2u : p.x_zero) + (unsigned)(((s & 3) ^ hc_swz_act(px)) * 16);
  }
  auto dma_act = [&](int it, int c0, int buf) {
    if (it * HC_T + tid < HC_ASLOTS) {
      _Float16* dst = s_act + buf * 2 * HC_ACT + (it * HC_T + wave * 64) * 8;
      const unsigned o = a_off[it] + (unsigned)c0 * 2u;
      hc_glds16(p.x_hi, o, dst);
      hc_glds16(p.x_lo, o, dst + HC_ACT);
    }
  };
  // weight fragment of (tile j, tap, chunk c0): row n0 + wq * 32 + j * 16 + fr, halves tap * C + c0 + kq * 8 .. + 7
  unsigned w_row[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wq * 32 + j * 16 + fr;
    w_row[j] = (n < p.N ? (unsigned)(n * 9 * p.C) * 2u : p.w_zero) + (unsigned)kq * 16u;
  }
  // buffer loads: resource descriptor in SGPRs, the lane's row offset in ONE VGPR per tile, (tap, chunk) as the wave-uniform
  // scalar offset - no per-tap 64-bit addresses in registers (plain pointers: 256 VGPRs + 10 spills)
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
  const __amdgpu_buffer_rsrc_t r_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w_hi), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w_lo), 0, -1, 0x00020000);
  auto load_w = [&](half8 (&wh)[2], half8 (&wl)[2], int tap, int c0) {
    const int so = (tap * p.C + c0) * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      wh[j] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r_hi, (int)w_row[j], so, 0));
      wl[j] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r_lo, (int)w_row[j], so, 0));
    }
  };

  f32x4 acc_m[8][2], acc_x[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc_m[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}, acc_x[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunks = p.C / HC_BK;
  half8 bh[2], bl[2], nh[2], nl[2];
  load_w(bh, bl, 0, 0);
#pragma unroll
  for (int it = 0; it < HC_AIT; ++it) dma_act(it, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * HC_BK;
    const _Float16* act = s_act + (ch & 1) * 2 * HC_ACT;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // the next step's weights (the loads stay in flight under this step's 48 MFMAs)
      if (tap < 8)
        load_w(nh, nl, tap + 1, c0);
      else if (ch + 1 < nchunks)
        load_w(nh, nl, 0, c0 + HC_BK);
      if (tap < HC_AIT && ch + 1 < nchunks) dma_act(tap, c0 + HC_BK, (ch + 1) & 1);   // next halo, one slot round per tap
      __builtin_amdgcn_sched_barrier(0);                 // (left to itself the compiler sinks the weight loads to the end of the step
                                                         //  and waits for them with vmcnt(0) at the top of the next one)
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int g = 0; g < 4; ++g) {                      // pairs of M-tiles: row 2 wr2 + (g >> 1), x = 32 (g & 1) + 16 i
        half8 ah[2], al[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int hp = (2 * wr2 + (g >> 1) + dy) * HC_HX + ((g & 1) * 2 + i) * 16 + dx + fr;
          const int ao = hp * HC_BK + ((kq ^ hc_swz_act(hp)) * 8);
          ah[i] = *reinterpret_cast<const half8*>(act + ao);
          al[i] = *reinterpret_cast<const half8*>(act + HC_ACT + ao);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc_m[2 * g + i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc_m[2 * g + i][j], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc_m[2 * g + i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc_x[2 * g + i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc_x[2 * g + i][j], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc_x[2 * g + i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc_x[2 * g + i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc_x[2 * g + i][j], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc_x[2 * g + i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // the weights of the next step (issued a full step ago) and this step's halo pieces: waited for HERE, explicitly - the DMA issue
      // sits in a conditional, and behind a control-flow merge the compiler's own wait insertion falls back to vmcnt(0) right
      // before the next step's first MFMA, i.e. directly behind that step's fresh loads
      __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0), lgkmcnt / expcnt untouched
#pragma unroll
      for (int j = 0; j < 2; ++j) bh[j] = nh[j], bl[j] = nl[j];
    }
    if (ch + 1 < nchunks) {      // the next chunk's halo landed; every wave is done reading this chunk's buffer
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue (the arithmetic of hc_epilogue on this wave tile: M-tile i = row 2 wr2 + (i >> 2), x = 16 (i & 3))
  const int e_a = ff3d_ld_exp(p.sc.a_exp);
  const float sc_in = ff3d_pow2(e_a + ff3d_ld_exp(p.sc.w_exp));
  float sc_out = 1.f;
  if (p.sc.out_exp) {
    const int e_out = ff3d_out_exp(p.sc, e_a, false, INFINITY);
    if (!p.out) sc_out = ff3d_pow2(-e_out);
    if (lid == 0 && tid == 0) *p.sc.out_exp = e_out;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int y = ty0 + 2 * wr2 + (i >> 2);
    if (y >= p.H) continue;
    if (TR) {    // pair output: lane = pixel (fr), channels n .. n + 3
      const int x = tx0 + (i & 3) * 16 + fr;
      if (x >= p.W) continue;
      const long long pix = ((long long)b * p.H + y) * p.W + x;
      const bool n4 = (p.N & 3) == 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wq * 32 + j * 16 + kq * 4;
        if (n >= p.N) continue;
        _Float16 h[4], l[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc_m[i][j][r] + acc_x[i][j][r] * (1.f / 2048.f), sc_in, (p.bias && n + r < p.N) ? p.bias[n + r] : 0.f);
          if (p.relu) v = fmaxf(v, 0.f);
          v *= sc_out;
          h[r] = (_Float16)v;
          l[r] = (_Float16)((v - (float)h[r]) * 2048.f);
        }
        const long long o = pix * p.N + n;
        if (n4) {
          *reinterpret_cast<uint2*>(p.out_hi + o) = *reinterpret_cast<uint2*>(h);
          *reinterpret_cast<uint2*>(p.out_lo + o) = *reinterpret_cast<uint2*>(l);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) p.out_hi[o + r] = h[r], p.out_lo[o + r] = l[r];
        }
      }
    } else {     // NCHW fp32: D row = 4 kq + r (pixel), col = fr (channel)
      const int x = tx0 + (i & 3) * 16 + kq * 4;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wq * 32 + j * 16 + fr;
        if (n >= p.N) continue;
        const float bj = p.bias ? p.bias[n] : 0.f;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fmaf(acc_m[i][j][r] + acc_x[i][j][r] * (1.f / 2048.f), sc_in, bj);
          if (p.relu) v[r] = fmaxf(v[r], 0.f);
        }
        float* o = p.out + (((long long)b * p.N + n) * p.H + y) * p.W + x;
        if (x + 3 < p.W && (p.W & 3) == 0) {
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (x + r < p.W) o[r] = v[r];
        }
      }
    }
  }
}
#endif  // FF3D_BUILD_EXPERIMENTS (wreg)

#ifdef FF3D_BUILD_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------------------
// Round 4 experiment (FF3D_HALO_M32=1): the 4 x 64 kernel on v_mfma_f32_32x32x16_f16.  The 16x16x32 form issues at ~17 cycles per
// instruction back to back (MI355X_MICROARCH.md: 16 would be the pipe's rate; the MFMA-only ablation of the kernel above reaches
// 2.06 PFLOP/s = 82 % of peak), the 32x32x16 form at exactly 32 cycles for twice the work (2 495 TFLOP/s in the micro-benchmark).
// Same block tile, DMA, LDS layout and swizzles; a wave's 64 pixels x 64 channels are 2 x 2 tiles of 32 x 32, a K-step of 32
// channels is two MFMA K-steps of 16: fragment of tile t, K-half s = row (pixel | channel) t * 32 + (lane & 31), 16-byte chunk
// 2 s + (lane >> 5) of the 64-byte LDS row - the same 16 fragment reads per (chunk, tap) step as above, 24 MFMAs instead of 48.
// (Round 1 measured a 32x32 tiling 8 % slower - before the pass-major order, the alignment-free halo swizzle and the per-tap weight
// stream of this kernel.)
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <bool TR>
__global__ __launch_bounds__(HC_T, 1) void conv3x3_halo_m32_f16x3_kernel(HaloParams p) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  _Float16* const s_act = lds;                           // [2 buffers][2 planes][HC_ACT]
  _Float16* const s_wt = lds + 2 * 2 * HC_ACT;           // [2 buffers][2 planes][HC_WT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, lh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_x = (p.W + HC_X - 1) / HC_X, tiles_y = (p.H + HC_Y - 1) / HC_Y, n_tiles = (p.N + HC_BN - 1) / HC_BN;
  const unsigned lid = ff3d_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = (int)(lid % n_tiles);
  const int sp = (int)(lid / n_tiles), b = sp / (tiles_x * tiles_y), t = sp % (tiles_x * tiles_y);
  const int ty0 = (t / tiles_x) * HC_Y, tx0 = (t % tiles_x) * HC_X, n0 = nt * HC_BN;

  unsigned a_off[HC_AIT];
#pragma unroll
  for (int it = 0; it < HC_AIT; ++it) {
    const int s = it * HC_T + tid, px = min(s >> 2, HC_HALO - 1), ly = px / HC_HX, lx = px - ly * HC_HX;
    const int gy = ty0 + ly - 1, gx = tx0 + lx - 1;
    const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    a_off[it] = (in ? (unsigned)(((b * p.H + gy) * p.W + gx) * p.C) * 2u : p.x_zero) + (unsigned)(((s & 3) ^ hc_swz_act(px)) * 16);
  }
  unsigned w_off;
  {
    const int row = tid >> 2, n = n0 + row;
    w_off = (n < p.N ? (unsigned)(n * 9 * p.C) * 2u : p.w_zero) + (unsigned)(((tid & 3) ^ hc_swz(row)) * 16);
  }
  auto dma_act = [&](int it, int c0, int buf) {
    if (it * HC_T + tid < HC_ASLOTS) {
      _Float16* dst = s_act + buf * 2 * HC_ACT + (it * HC_T + wave * 64) * 8;
      const unsigned o = a_off[it] + (unsigned)c0 * 2u;
      hc_glds16(p.x_hi, o, dst);
      hc_glds16(p.x_lo, o, dst + HC_ACT);
    }
  };
  auto dma_wt = [&](int tap, int c0, int buf) {
    _Float16* dst = s_wt + buf * 2 * HC_WT + (wave * 64) * 8;
    const unsigned o = w_off + (unsigned)(tap * p.C + c0) * 2u;
    hc_glds16(p.w_hi, o, dst);
    hc_glds16(p.w_lo, o, dst + HC_WT);
  };

  f32x16 acc_m[2][2], acc_x[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_m[i][j][r] = 0.f, acc_x[i][j][r] = 0.f;

  int b_rd[2][2];                                        // weight fragment offsets [tile j][K-half s] (tap invariant)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rb = wc * 64 + j * 32 + l32;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) b_rd[j][s2] = rb * HC_BK + (((2 * s2 + lh) ^ hc_swz(rb)) * 8);
  }

  const int nchunks = p.C / HC_BK;
#pragma unroll
  for (int it = 0; it < HC_AIT; ++it) dma_act(it, 0, 0);
  dma_wt(0, 0, 0);
  int wbuf = 0;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * HC_BK;
    const _Float16* act = s_act + (ch & 1) * 2 * HC_ACT;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tap < 8)
        dma_wt(tap + 1, c0, wbuf ^ 1);
      else if (ch + 1 < nchunks)
        dma_wt(0, c0 + HC_BK, wbuf ^ 1);
      if (tap < HC_AIT && ch + 1 < nchunks) dma_act(tap, c0 + HC_BK, (ch + 1) & 1);
      const int dy = tap / 3, dx = tap - dy * 3;
      const _Float16* wt = s_wt + wbuf * 2 * HC_WT;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {                   // the two 16-channel halves of the K-step
        half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int hp = (wr + dy) * HC_HX + i * 32 + dx + l32;
          const int ao = hp * HC_BK + (((2 * s2 + lh) ^ hc_swz_act(hp)) * 8);
          ah[i] = *reinterpret_cast<const half8*>(act + ao);
          al[i] = *reinterpret_cast<const half8*>(act + HC_ACT + ao);
          bh[i] = *reinterpret_cast<const half8*>(wt + b_rd[i][s2]);
          bl[i] = *reinterpret_cast<const half8*>(wt + HC_WT + b_rd[i][s2]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc_m[i][j] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc_m[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc_m[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc_x[i][j] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc_x[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc_x[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc_x[i][j] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc_x[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc_x[i][j], 0, 0, 0);
      }
      wbuf ^= 1;
    }
  }

  // ---- epilogue.  C / D of the 32 x 32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const int e_a = ff3d_ld_exp(p.sc.a_exp);
  const float sc_in = ff3d_pow2(e_a + ff3d_ld_exp(p.sc.w_exp));
  float sc_out = 1.f;
  if (p.sc.out_exp) {
    const int e_out = ff3d_out_exp(p.sc, e_a, false, INFINITY);
    if (!p.out) sc_out = ff3d_pow2(-e_out);
    if (lid == 0 && tid == 0) *p.sc.out_exp = e_out;
  }
  const int y = ty0 + wr;
  if (y >= p.H) return;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (TR) {    // D[channel][pixel]: lane = pixel l32 of M-tile i; channels 8 g + 4 lh .. + 3 of N-tile j
      const int x = tx0 + i * 32 + l32;
      if (x >= p.W) continue;
      const long long pix = ((long long)b * p.H + y) * p.W + x;
      const bool n4 = (p.N & 3) == 0;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wc * 64 + j * 32 + 8 * g + 4 * lh;
          if (n >= p.N) continue;
          _Float16 h[4], l[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = fmaf(acc_m[i][j][4 * g + r] + acc_x[i][j][4 * g + r] * (1.f / 2048.f), sc_in,
                           (p.bias && n + r < p.N) ? p.bias[n + r] : 0.f);
            if (p.relu) v = fmaxf(v, 0.f);
            v *= sc_out;
            h[r] = (_Float16)v;
            l[r] = (_Float16)((v - (float)h[r]) * 2048.f);
          }
          const long long o = pix * p.N + n;
          if (n4) {
            *reinterpret_cast<uint2*>(p.out_hi + o) = *reinterpret_cast<uint2*>(h);
            *reinterpret_cast<uint2*>(p.out_lo + o) = *reinterpret_cast<uint2*>(l);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) p.out_hi[o + r] = h[r], p.out_lo[o + r] = l[r];
          }
        }
    } else {     // D[pixel][channel]: lane = channel l32 of N-tile j; pixels 8 g + 4 lh .. + 3 of M-tile i
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wc * 64 + j * 32 + l32;
        if (n >= p.N) continue;
        const float bj = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int x = tx0 + i * 32 + 8 * g + 4 * lh;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = fmaf(acc_m[i][j][4 * g + r] + acc_x[i][j][4 * g + r] * (1.f / 2048.f), sc_in, bj);
            if (p.relu) v[r] = fmaxf(v[r], 0.f);
          }
          float* o = p.out + (((long long)b * p.N + n) * p.H + y) * p.W + x;
          if (x + 3 < p.W && (p.W & 3) == 0) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (x + r < p.W) o[r] = v[r];
          }
        }
      }
    }
  }
}
#endif  // FF3D_BUILD_EXPERIMENTS (m32)

#ifdef FF3D_BUILD_EXPERIMENTS   // measured-slower variants kept as evidence (profiles/r03_m_*, r03_n_*, r03_f_*): python -m focalformer3d_amd.build with FF3D_BUILD_EXPERIMENTS=1
// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the software-pipelined form of the 4 x 64 kernel above (same tile, same arithmetic, same results).
// What the ISA of the kernel above shows (and of every kernel of rounds 1-2): the compiler sinks each ds_read_b128 of a fragment
// to just before its first MFMA and follows it with s_waitcnt lgkmcnt(0) - 7 exposed LDS round trips per 48-MFMA step, the
// "~55 % MFMA-busy whatever the tile / ring depth" plateau of splitmm.hip's header.  The hardware can overlap them; the
// schedule has to be written down:
//   * fragment reads are inline-asm ds_read_b128 (invisible to the compiler's waitcnt insertion), every s_waitcnt lgkmcnt(n)
//     is counted by hand and tied to the registers it guards, __builtin_amdgcn_sched_barrier(0) pins reads / passes in place;
//   * a step t (one filter tap of one 32-channel chunk) runs
//         issue bl(t), al(t) | pass 1: x_hi * w_hi  (its operands were fetched during step t - 1) | lgkmcnt(4): pass 2: x_hi * w_lo'
//         | lgkmcnt(0), vmcnt, s_barrier = S_t | DMA issue | issue x_hi(t + 1), w_hi(t + 1) | pass 3: x_lo' * w_hi
//     so no MFMA ever waits for a read issued less than 16 MFMAs (256 cycles) earlier;
//   * the weight ring has THREE stages: S_t publishes stage t + 1 (needed by the fetch that follows it), the DMA issued after S_t
//     refills the stage step t has just finished with weights t + 3 - two full steps of flight time instead of one, with a counted
//     vmcnt (the pieces issued after S_(t-1) stay in flight across S_t);
//   * three weight-fragment register sets rotate with period 3 (w_hi(t) in set t % 3, w_lo'(t) in set (t + 2) % 3, the fetch of
//     w_hi(t + 1) goes to set (t + 1) % 3): 9 taps = 3 periods, so every index is static in the unrolled tap loop.
// LDS: halo 99 KiB + 3 x 16 KiB weights = 147 KiB.
constexpr size_t HP_LDS_BYTES = (size_t)(2 * 2 * HC_ACT + 3 * 2 * HC_WT) * sizeof(_Float16);

#define HP_LDS_READ(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))
#define HP_WAIT_LGKM(n, a) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::"memory")

// MFMAs as inline asm with the accumulator tied to an AGPR ("+a"): updated in place, 128 AGPRs for the 32 accumulators.  (With
// the builtin the register allocator kept the accumulators in VGPRs with vdst != srcC and spilled accumulators inside the loop.)
// x = activation fragment (16 pixels x 32 k), w = weight fragment (16 channels x 32 k); TR swaps the operands (D^T).
#define HP_MFMA(acc, x, w)                                                                              \
  do {                                                                                                  \
    if (TR)                                                                                             \
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(x));             \
    else                                                                                                \
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(w));             \
  } while (0)

// VAR bit 0: the two waves of a SIMD (w, w + 4) issue their DMA pieces at different points of the step - waves 0-3 right after
//            S_t, waves 4-7 after pass 3 (an LDS-DMA piece costs its wave 60 - 185 issue cycles, MI355X_MICROARCH.md: while
//            one wave of the SIMD issues pieces the other one's MFMAs keep the pipe busy; the VM-queue order, hence every
//            counted wait, is unchanged);
//     bit 1, bit 2: tuning ablations (WRONG results): no DMA inside the loop; no s_barrier inside the loop.
template <bool TR, int VAR>
__global__ __launch_bounds__(HC_T, 1) void conv3x3_halo_pipe_f16x3_kernel(HaloParams p) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  _Float16* const s_act = lds;                           // [2 buffers][2 planes][HC_ACT]
  _Float16* const s_wt = lds + 2 * 2 * HC_ACT;           // [3 stages][2 planes][HC_WT]
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  constexpr unsigned ACT_PLANE_B = HC_ACT * 2, ACT_BUF_B = 2 * ACT_PLANE_B, WT_PLANE_B = HC_WT * 2, WT_STAGE_B = 2 * WT_PLANE_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_x = (p.W + HC_X - 1) / HC_X, tiles_y = (p.H + HC_Y - 1) / HC_Y, n_tiles = (p.N + HC_BN - 1) / HC_BN;
  const unsigned lid = ff3d_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = (int)(lid % n_tiles);
  const int sp = (int)(lid / n_tiles), b = sp / (tiles_x * tiles_y), t = sp % (tiles_x * tiles_y);
  const int ty0 = (t / tiles_x) * HC_Y, tx0 = (t % tiles_x) * HC_X, n0 = nt * HC_BN;

  // ---- DMA slot geometry (as above)
  unsigned a_off[HC_AIT];
#pragma unroll
  for (int it = 0; it < HC_AIT; ++it) {
    const int s = it * HC_T + tid, px = min(s >> 2, HC_HALO - 1), ly = px / HC_HX, lx = px - ly * HC_HX;
    const int gy = ty0 + ly - 1, gx = tx0 + lx - 1;
    const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    a_off[it] = (in ? (unsigned)(((b * p.H + gy) * p.W + gx) * p.C) * 2u : p.x_zero) + (unsigned)(((s & 3) ^ hc_swz_act(px)) * 16);
  }
  unsigned w_off;
  {
    const int row = tid >> 2, n = n0 + row;
    w_off = (n < p.N ? (unsigned)(n * 9 * p.C) * 2u : p.w_zero) + (unsigned)(((tid & 3) ^ hc_swz(row)) * 16);
  }
  // wave-uniform: does this wave issue pieces in slot round `it` of the halo (rounds 0-2: every wave, round 3: wave 0 only)
  auto act_round_live = [&](int it) { return it * HC_T + wave * 64 < HC_ASLOTS; };
  auto dma_act = [&](int it, int c0, int buf) {
    if (it * HC_T + tid < HC_ASLOTS) {
      _Float16* dst = s_act + buf * 2 * HC_ACT + (it * HC_T + wave * 64) * 8;
      const unsigned o = a_off[it] + (unsigned)c0 * 2u;
      hc_glds16(p.x_hi, o, dst);
      hc_glds16(p.x_lo, o, dst + HC_ACT);
    }
  };
  auto dma_wt = [&](int tap, int c0, int stage) {
    _Float16* dst = s_wt + stage * 2 * HC_WT + (wave * 64) * 8;
    const unsigned o = w_off + (unsigned)(tap * p.C + c0) * 2u;
    hc_glds16(p.w_hi, o, dst);
    hc_glds16(p.w_lo, o, dst + HC_WT);
  };

  f32x4 acc_m[4][4], acc_x[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc_m[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}, acc_x[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addresses: a per-lane base register + an IMMEDIATE offset per (tap, tile, plane, stage), so that nothing
  //      tap-dependent lives in registers (the first version kept ~70 loop-invariant addresses and spilled).
  //      weights:     wb[j] + stage * WT_STAGE_B + plane * WT_PLANE_B
  //      activations: halo pixel hp = hp0 + c, c = dy * 66 + dx + i * 16; byte address hp * 64 + (kq * 16 ^ swizzle(hp)), and
  //                   the swizzle bit (bit 2 of hp) depends on c only through r = c & 7 = 2 dy + dx (0..6): seven per-lane bases
  //                   xa[r] = lds0 + hp0 * 64 + (kq * 16 ^ bit2(hp0 + r) * 32), offset c * 64 + plane * ACT_PLANE_B; the halo buffer
  //                   of the chunk is added to the bases once per chunk.
  unsigned wb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rb = wc * 64 + j * 16 + fr;
    wb[j] = lds0 + 2 * ACT_BUF_B + (unsigned)(rb * HC_BK + ((kq ^ hc_swz(rb)) * 8)) * 2u;
  }
  const unsigned hp0 = (unsigned)(wr * HC_HX + fr);
  unsigned xa[7];                                        // (running: toggled between the two halo buffers at every chunk end)
#pragma unroll
  for (int r = 0; r < 7; ++r) xa[r] = lds0 + hp0 * 64u + (((unsigned)kq * 16u) ^ ((((hp0 + r) >> 2) & 1u) * 32u));

  const int nchunks = p.C / HC_BK, T = nchunks * 9;
  // ---- prologue: halo of chunk 0, weights of steps 0, 1, 2; operands of pass 1 of step 0
#pragma unroll
  for (int it = 0; it < HC_AIT; ++it) dma_act(it, 0, 0);
  dma_wt(0, 0, 0);
  dma_wt(1, 0, 1);
  dma_wt(2, 0, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  half8 ah[4], al[4], wf[3][4];                          // wf: the three rotating weight-fragment sets
#pragma unroll
  for (int i = 0; i < 4; ++i) HP_LDS_READ(ah[i], xa[0], i * 16 * 64);
#pragma unroll
  for (int j = 0; j < 4; ++j) HP_LDS_READ(wf[0][j], wb[j], 0);

  int step = 0;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * HC_BK;
    const unsigned flip = (ch & 1) ? 0u - ACT_BUF_B : ACT_BUF_B;   // to the other halo buffer (wave-uniform)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap, ++step) {
      // sets: w_hi(t) in H = tap % 3, w_lo'(t) -> L = (tap + 2) % 3, fetch of w_hi(t + 1) -> N = (tap + 1) % 3; the weight
      // stage of step t is t % 3 = tap % 3 (9 % 3 == 0)
      half8(&bh)[4] = wf[tap % 3];
      half8(&bl)[4] = wf[(tap + 2) % 3];
      half8(&bn)[4] = wf[(tap + 1) % 3];
      const int dy = tap / 3, dx = tap - dy * 3;
      // (the DMA source offsets are recomputed per step from these five registers: left to itself the compiler keeps one
      // pre-added copy per tap and spills them)
      asm volatile("" : "+v"(w_off), "+v"(a_off[0]), "+v"(a_off[1]), "+v"(a_off[2]), "+v"(a_off[3]));
#pragma unroll
      for (int j = 0; j < 4; ++j) HP_LDS_READ(bl[j], wb[j], (tap % 3) * WT_STAGE_B + WT_PLANE_B);
#pragma unroll
      for (int i = 0; i < 4; ++i) HP_LDS_READ(al[i], xa[2 * dy + dx], (dy * HC_HX + dx + i * 16) * 64 + ACT_PLANE_B);
      __builtin_amdgcn_sched_barrier(0);
      HP_WAIT_LGKM(8, ah);                               // x_hi(t), w_hi(t): fetched before the 8 reads just issued
      HP_WAIT_LGKM(8, bh);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          HP_MFMA(acc_m[i][j], ah[i], bh[j]);
      __builtin_amdgcn_sched_barrier(0);
      HP_WAIT_LGKM(4, bl);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          HP_MFMA(acc_x[i][j], ah[i], bl[j]);
      __builtin_amdgcn_sched_barrier(0);
      HP_WAIT_LGKM(0, al);
      // ---- S_t: the weights of step t + 1 have landed (behind them in the VM queue: the halo pieces and the weights t + 2
      //      issued after S_(t-1)); every wave's reads of this step's stage / of this chunk's last use of the halo are retired
      {
        const int ptap = tap == 0 ? 8 : tap - 1;                                   // the tap of step t - 1
        const bool prev_act = step > 0 && ptap < HC_AIT && (tap == 0 ? ch : ch + 1) < nchunks && act_round_live(ptap);
        const int behind = (prev_act ? 2 : 0) + ((step > 0 && step + 2 < T) ? 2 : (step == 0 ? 2 : 0));
        if (behind == 4)
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (behind == 2)
          asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (!(VAR & 4)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      auto issue_dma = [&]() {
        if (VAR & 2) return;
        if (tap < HC_AIT && ch + 1 < nchunks) dma_act(tap, c0 + HC_BK, (ch + 1) & 1);   // next halo, one slot round per tap
        if (step + 3 < T) {                                                             // weights t + 3 -> the stage of step t
          const int t3 = tap + 3;
          dma_wt(t3 < 9 ? t3 : t3 - 9, t3 < 9 ? c0 : c0 + HC_BK, tap % 3);
        }
      };
      if (!(VAR & 1) || wave < 4) issue_dma();
      if (step + 1 < T) {                                                             // pass-1 operands of step t + 1
#pragma unroll
        for (int j = 0; j < 4; ++j) HP_LDS_READ(bn[j], wb[j], ((tap + 1) % 3) * WT_STAGE_B);
        if (tap == 8) {                                  // tap 0 of the next chunk (r = 0) in the other halo buffer
          const unsigned xn0 = xa[0] + flip;
#pragma unroll
          for (int i = 0; i < 4; ++i) HP_LDS_READ(ah[i], xn0, i * 16 * 64);
        } else {
          const int ndy = (tap + 1) / 3, ndx = (tap + 1) - ndy * 3;
#pragma unroll
          for (int i = 0; i < 4; ++i) HP_LDS_READ(ah[i], xa[2 * ndy + ndx], (ndy * HC_HX + ndx + i * 16) * 64);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          HP_MFMA(acc_x[i][j], al[i], bh[j]);
      __builtin_amdgcn_sched_barrier(0);
      if ((VAR & 1) && wave >= 4) issue_dma();
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 7; ++r) xa[r] += flip;
  }
  // the accumulators were written by inline-asm MFMAs the compiler's hazard recogniser does not see: XDL write -> VALU read of
  // the same registers needs up to 19 wait states (8-pass MFMA: 11)
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  hc_epilogue<TR>(p, acc_m, acc_x, lid, tid, b, ty0, tx0, n0, wr, wc, fr, kq, lane);
}
#undef HP_LDS_READ
#undef HP_MFMA
#undef HP_WAIT_LGKM


// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the 8 x 64-pixel form.  The 4 x 64 kernel above spends 0.8 of its 2.95 ms on the L2 / fabric side of the 13.6 GB its
// 8640 blocks pull per launch, 10.2 GB of it the 1.2 MB of weights re-streamed by every block (ablations in the header).  Here
// a block owns 8 x 64 pixels x 128 output channels, so the same weight stream feeds twice the MFMAs (5.1 GB per launch) and
// the halo shrinks from 1.55 x to 1.29 x of the tile:
//   * ONE accumulator per output tile instead of two.  The cross terms hi * lo' carry a factor 2^-11; instead of a second
//     accumulator scaled in the epilogue, the ACTIVATION fragment of each cross product is pre-scaled by 2^-11 in registers
//     (x_hi * w_lo' -> (x_hi 2^-11) * w_lo', x_lo' * w_hi -> (x_lo' 2^-11) * w_hi; v_pk_mul_f16: exact unless the scaled value
//     drops below fp16's normal range, i.e. below 2^-3 in a tensor normalised to 2^14 - an absolute error <= 2^-25 * 2^14 per
//     product against main terms of up to 2^28), so main and cross terms add into the same fp32 accumulator: 128 accumulator
//     registers for a 128-pixel x 64-channel wave tile.
//   * waves: 8 = 4 (row pairs) x 2 (64-channel halves); wave tile = 2 rows x 64 pixels x 64 channels = 8 x 4 MFMA tiles;
//     per K-step 24 fragment reads for 96 MFMAs (0.25 per MFMA; 0.33 above).
//   * the K dimension of the implicit GEMM is cut in 16-channel HALF chunks: a K-step of 32 = 2 (tap, half-chunk) units.  A
//     32-channel chunk is 18 units = 9 steps: (h0: taps 01 | 23 | 45 | 67), (h0 tap 8 + h1 tap 0), (h1: taps 12 | 34 | 56 | 78).
//     The 10 x 66 halo of one half chunk is 41.25 KiB for both planes; TWO of them form a ring (82.5 KiB) - the double-buffered
//     32-channel halo of this tile (165 KiB) would not fit: half 0 of the next chunk is DMAed during steps 5-7 (half 0 is last
//     read in step 4), half 1 during steps 0-2 of the next chunk (needed from its step 4).  No bubble at chunk boundaries.
//   * LDS: halo ring 82.5 KiB + weights [2][plane][128 n][32 k] 32 KiB = 114.5 KiB, one block per CU.  Halo rows are 32 B (two
//     16-byte chunks per pixel): a ds_read_b128 service group (lanes {0-3, 12-15, 20-27}) reads pixels fr = 0-3, 12-15 chunk 0
//     and pixels 4-11 chunk 1 = 16 distinct 16-byte bank columns without any swizzle.
constexpr int H8_Y = 8, H8_X = 64, H8_HX = H8_X + 2, H8_HALO = (H8_Y + 2) * H8_HX;        // 660 halo pixels
constexpr int H8_HK = 16;                                                               // channels per half chunk
constexpr int H8_SLOT = H8_HALO * H8_HK;                                                // halves per plane and ring slot
constexpr int H8_ASLOTS = H8_HALO * 2;                                                  // 16-byte pieces per plane and slot
constexpr int H8_AIT = (H8_ASLOTS + HC_T - 1) / HC_T;                                   // 3 rounds
constexpr size_t H8_LDS_BYTES = (size_t)(2 * 2 * H8_SLOT + 2 * 2 * HC_WT) * sizeof(_Float16);

template <bool TR>
__global__ __launch_bounds__(HC_T, 1) void conv3x3_halo8_f16x3_kernel(HaloParams p) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  _Float16* const s_act = lds;                           // [2 ring slots][2 planes][H8_SLOT]
  _Float16* const s_wt = lds + 2 * 2 * H8_SLOT;          // [2 buffers][2 planes][HC_WT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_x = (p.W + H8_X - 1) / H8_X, tiles_y = (p.H + H8_Y - 1) / H8_Y, n_tiles = (p.N + HC_BN - 1) / HC_BN;
  const unsigned lid = ff3d_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = (int)(lid % n_tiles);
  const int sp = (int)(lid / n_tiles), b = sp / (tiles_x * tiles_y), t = sp % (tiles_x * tiles_y);
  const int ty0 = (t / tiles_x) * H8_Y, tx0 = (t % tiles_x) * H8_X, n0 = nt * HC_BN;

  // ---- DMA geometry.  Halo: piece s = it * 512 + tid of a slot: pixel s >> 1, 16-byte chunk s & 1 (channels 8 (s & 1) ..)
  unsigned a_off[H8_AIT];
#pragma unroll
  for (int it = 0; it < H8_AIT; ++it) {
    const int s = it * HC_T + tid, px = min(s >> 1, H8_HALO - 1), ly = px / H8_HX, lx = px - ly * H8_HX;
    const int gy = ty0 + ly - 1, gx = tx0 + lx - 1;
    const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    a_off[it] = (in ? (unsigned)(((b * p.H + gy) * p.W + gx) * p.C) * 2u : p.x_zero) + (unsigned)((s & 1) * 16);
  }
  // Weights: thread = (row tid >> 2, LDS chunk position tid & 3); the logical chunk lc = position ^ swizzle(row) holds unit
  // lc >> 1 (first / second (tap, half chunk) of the K-step), channels 8 (lc & 1) .. of that unit
  const int w_row = tid >> 2, w_lc = (tid & 3) ^ hc_swz(w_row);
  const unsigned w_base = (n0 + w_row < p.N ? (unsigned)((n0 + w_row) * 9 * p.C) * 2u : p.w_zero) + (unsigned)((w_lc & 1) * 16);
  const bool w_real = n0 + w_row < p.N;
  auto dma_act = [&](int it, int c_half, int slot) {     // one round of the halo of the 16 channels c_half .. c_half + 15
    if (it * HC_T + tid < H8_ASLOTS) {
      _Float16* dst = s_act + slot * 2 * H8_SLOT + (it * HC_T + wave * 64) * 8;   // wave-uniform; the DMA adds lane * 16 B
      const unsigned o = a_off[it] + (unsigned)c_half * 2u;
      hc_glds16(p.x_hi, o, dst);
      hc_glds16(p.x_lo, o, dst + H8_SLOT);
    }
  };
  auto dma_wt = [&](int step, int c0, int buf) {         // K-step `step` (0..8) of the 32-channel chunk at c0
    const int u = 2 * step + (w_lc >> 1), half = u >= 9 ? 1 : 0, tap = u - 9 * half;
    _Float16* dst = s_wt + buf * 2 * HC_WT + (wave * 64) * 8;
    const unsigned o = w_base + (w_real ? (unsigned)(tap * p.C + c0 + half * H8_HK) * 2u : 0u);
    hc_glds16(p.w_hi, o, dst);
    hc_glds16(p.w_lo, o, dst + HC_WT);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int b_rd[4];                                           // weight fragment offsets (step invariant)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rb = wc * 64 + j * 16 + fr;
    b_rd[j] = rb * HC_BK + ((kq ^ hc_swz(rb)) * 8);
  }
  // halo offset (halves) of this lane for M-tile 0 without the tap shift: pixel (2 wr) * 66 + fr, chunk kq & 1
  const int a_rd0 = ((2 * wr) * H8_HX + fr) * H8_HK + (kq & 1) * 8;
  const _Float16 k_lo = (_Float16)(1.f / 2048.f);

  const int nchunks = p.C / HC_BK;
#pragma unroll
  for (int it = 0; it < H8_AIT; ++it) dma_act(it, 0, 0), dma_act(it, H8_HK, 1);
  dma_wt(0, 0, 0);
  int wbuf = 0;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * HC_BK;
#pragma unroll
    for (int st = 0; st < 9; ++st) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();           // this step's weights and every halo piece issued so far landed; previous reads retired
      if (st < 8)
        dma_wt(st + 1, c0, wbuf ^ 1);
      else if (ch + 1 < nchunks)
        dma_wt(0, c0 + HC_BK, wbuf ^ 1);
      if (ch + 1 < nchunks) {
        if (st >= 5 && st < 5 + H8_AIT) dma_act(st - 5, c0 + HC_BK, 0);              // half 0 of the next chunk
      }
      if (ch > 0 && st < H8_AIT) dma_act(st, c0 + H8_HK, 1);                          // half 1 of THIS chunk (needed from step 4)
      // this lane's unit of the step: lanes kq 0, 1 -> unit 2 st, lanes kq 2, 3 -> unit 2 st + 1
      const int u = 2 * st + (kq >> 1), half = u >= 9 ? 1 : 0, tap = u - 9 * half;
      const int dy = tap / 3, dx = tap - dy * 3;
      const _Float16* act = s_act + half * 2 * H8_SLOT + a_rd0 + (dy * H8_HX + dx) * H8_HK;
      const _Float16* wt = s_wt + wbuf * 2 * HC_WT;
      half8 bh[4], bl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bh[j] = *reinterpret_cast<const half8*>(wt + b_rd[j]);
        bl[j] = *reinterpret_cast<const half8*>(wt + HC_WT + b_rd[j]);
      }
      // PASS-MAJOR order: the three MFMAs of one output tile (hi*hi, the two cross terms) depend on each other through the
      // accumulator, and a v_mfma_f32_16x16x32_f16 issued right behind the one that produces its C operand waits out the
      // full 8-pass latency (~32 cycles instead of the 16-cycle issue interval).  Tile-major order - three dependent MFMAs
      // back to back, what every split-fp16 kernel of rounds 1-2 did - caps the pipe at 48 / 80 = 60 % (the "56 % MFMA-busy
      // whatever the tile shape" of splitmm.hip's header).  Here the 16 tiles of a 4-tile row group take pass 0, then pass 1,
      // then pass 2: dependent instructions are 16 MFMAs apart.
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        half8 ah[4], al[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ao = (g * H8_HX + i * 16) * H8_HK;
          ah[i] = *reinterpret_cast<const half8*>(act + ao);
          al[i] = *reinterpret_cast<const half8*>(act + H8_SLOT + ao);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[4 * g + i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[4 * g + i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[4 * g + i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const half8 as = ah[i] * k_lo;                 // both cross terms take their 2^-11 on the activation side
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[4 * g + i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], as, acc[4 * g + i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_16x16x32_f16(as, bl[j], acc[4 * g + i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const half8 als = al[i] * k_lo;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[4 * g + i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], als, acc[4 * g + i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_16x16x32_f16(als, bh[j], acc[4 * g + i][j], 0, 0, 0);
        }
      }
      wbuf ^= 1;
    }
  }

  // ---- epilogue (as the 4 x 64 kernel's, one accumulator)
  const int e_a = ff3d_ld_exp(p.sc.a_exp);
  const float sc_in = ff3d_pow2(e_a + ff3d_ld_exp(p.sc.w_exp));
  float sc_out = 1.f;
  if (p.sc.out_exp) {
    const int e_out = ff3d_out_exp(p.sc, e_a, false, INFINITY);
    if (!p.out) sc_out = ff3d_pow2(-e_out);
    if (lid == 0 && tid == 0) *p.sc.out_exp = e_out;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int y = ty0 + 2 * wr + (i >> 2);
    if (y >= p.H) continue;
    if (TR) {   // pair output: lane = pixel x (column fr of the transposed tile), channels n .. n + 3
      const int x = tx0 + (i & 3) * 16 + fr;
      if (x >= p.W) continue;
      const long long pix = ((long long)b * p.H + y) * p.W + x;
      const bool n4 = (p.N & 3) == 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + kq * 4;
        if (n >= p.N) continue;
        _Float16 h[4], l[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc[i][j][r], sc_in, (p.bias && n + r < p.N) ? p.bias[n + r] : 0.f);
          if (p.relu) v = fmaxf(v, 0.f);
          v *= sc_out;
          h[r] = (_Float16)v;
          l[r] = (_Float16)((v - (float)h[r]) * 2048.f);
        }
        const long long o = pix * p.N + n;
        if (n4) {
          *reinterpret_cast<uint2*>(p.out_hi + o) = *reinterpret_cast<uint2*>(h);
          *reinterpret_cast<uint2*>(p.out_lo + o) = *reinterpret_cast<uint2*>(l);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) p.out_hi[o + r] = h[r], p.out_lo[o + r] = l[r];
        }
      }
    } else {    // NCHW fp32: D row = 4*kq + r (pixel x offset inside the M-tile), col = fr (output channel)
      const int x = tx0 + (i & 3) * 16 + kq * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + fr;
        if (n >= p.N) continue;
        const float bj = p.bias ? p.bias[n] : 0.f;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fmaf(acc[i][j][r], sc_in, bj);
          if (p.relu) v[r] = fmaxf(v[r], 0.f);
        }
        float* o = p.out + (((long long)b * p.N + n) * p.H + y) * p.W + x;
        if (x + 3 < p.W && (p.W & 3) == 0) {
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (x + r < p.W) o[r] = v[r];
        }
      }
    }
  }
}

#endif  // FF3D_BUILD_EXPERIMENTS

}  // namespace

// Grouped form: n (<= 4) convolutions of one shape in one launch.  Arrays of n device pointers / scale records on the HOST.
extern "C" int ff3d_conv3x3_halo_f16x3_group(int n, const void* const* x_hi, const void* const* x_lo, const void* const* w_hi,
                                             const void* const* w_lo, const float* const* bias, int apply_relu,
                                             float* const* out, void* const* out_hi, void* const* out_lo, int B, int C, int H,
                                             int W, int N, const ff3d_scale_t* const* scale_host, ff3d_stream_t stream) {
  FF3D_REQUIRE(n >= 1 && n <= HC_MAX_GROUP && x_hi && x_lo && w_hi && w_lo, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && C > 0 && C % HC_BK == 0 && H > 0 && W > 0 && N > 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(((long long)B * H * W + 1) * C * 2 < (1ll << 32) && ((long long)N + 1) * 9 * C * 2 < (1ll << 32),
               FF3D_ERR_BAD_SHAPE);
  const bool pair = out_hi && out_hi[0];
  FF3D_REQUIRE(!pair || N % 2 == 0, FF3D_ERR_BAD_SHAPE);
  const long long pad0 = (long long)((H + 3) / 4 * 4) * ((W + 63) / 64 * 64), pad1 = (long long)((H + 7) / 8 * 8) * ((W + 31) / 32 * 32);
  const bool geo1 = pad1 * 100 < pad0 * 99;      // (as in halo_conv_launch)
  const long long per = geo1 ? (long long)B * ((H + 7) / 8) * ((W + 31) / 32) * ((N + HC_BN - 1) / HC_BN)
                             : (long long)B * ((H + 3) / 4) * ((W + 63) / 64) * ((N + HC_BN - 1) / HC_BN);
  FF3D_REQUIRE(per * n < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  HaloGroup gp{};
  gp.per = (unsigned)per;
  for (int g = 0; g < n; ++g) {
    FF3D_REQUIRE(x_hi[g] && x_lo[g] && w_hi[g] && w_lo[g], FF3D_ERR_NULL);
    FF3D_REQUIRE(pair ? (out_hi[g] && out_lo && out_lo[g]) : (out && out[g]), FF3D_ERR_NULL);
    const ff3d_scale_t* sh = scale_host ? scale_host[g] : nullptr;
    FF3D_REQUIRE(!sh || !sh->out_exp || sh->w_bound, FF3D_ERR_NULL);
    gp.p[g] = HaloParams{static_cast<const _Float16*>(x_hi[g]), static_cast<const _Float16*>(x_lo[g]),
                         static_cast<const _Float16*>(w_hi[g]), static_cast<const _Float16*>(w_lo[g]), bias ? bias[g] : nullptr,
                         pair ? nullptr : out[g], pair ? static_cast<_Float16*>(out_hi[g]) : nullptr,
                         pair ? static_cast<_Float16*>(out_lo[g]) : nullptr, B, C, H, W, N, apply_relu ? 1 : 0,
                         (unsigned)((long long)B * H * W * C * 2), (unsigned)((long long)N * 9 * C * 2), ff3d_scale_from(sh)};
  }
  int dev = 0;
  (void)hipGetDevice(&dev);
  ff3d_clear_error();
  const dim3 grid((unsigned)(per * n));
#define FF3D_GROUP(TRV, GEOV)                                                                                               \
  do {                                                                                                                      \
    static bool configured[64] = {};                                                                                        \
    if (!configured[dev & 63]) {                                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_group_kernel<TRV, GEOV>),                         \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)HcGeo<GEOV>::LDS_BYTES) != hipSuccess)       \
        return FF3D_ERR_LAUNCH;                                                                                             \
      configured[dev & 63] = true;                                                                                          \
    }                                                                                                                       \
    hipLaunchKernelGGL((conv3x3_halo_group_kernel<TRV, GEOV>), grid, dim3(HC_T), HcGeo<GEOV>::LDS_BYTES,                    \
                       static_cast<hipStream_t>(stream), gp);                                                               \
  } while (0)
  static const bool tap2 = [] {
    const char* e = getenv("FF3D_HALO_TAP2");
    return !(e && e[0] == '0');
  }();
#define FF3D_GROUP_TAP2(TRV)                                                                                                \
  do {                                                                                                                      \
    static bool configured[64] = {};                                                                                        \
    if (!configured[dev & 63]) {                                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_group_tap2_kernel<TRV, 1>),                       \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)hc_tap2_lds_bytes<1>()) != hipSuccess)       \
        return FF3D_ERR_LAUNCH;                                                                                             \
      configured[dev & 63] = true;                                                                                          \
    }                                                                                                                       \
    hipLaunchKernelGGL((conv3x3_halo_group_tap2_kernel<TRV, 1>), grid, dim3(HC_T), hc_tap2_lds_bytes<1>(),                  \
                       static_cast<hipStream_t>(stream), gp);                                                               \
  } while (0)
  if (geo1 && tap2) {
    if (pair)
      FF3D_GROUP_TAP2(true);
    else
      FF3D_GROUP_TAP2(false);
  } else if (pair && geo1)
    FF3D_GROUP(true, 1);
  else if (pair)
    FF3D_GROUP(true, 0);
  else if (geo1)
    FF3D_GROUP(false, 1);
  else
    FF3D_GROUP(false, 0);
#undef FF3D_GROUP
#undef FF3D_GROUP_TAP2
  return ff3d_launch_status();
}

// Returns FF3D_ERR_UNSUPPORTED for shapes this form does not take (the caller then uses the implicit GEMM).
// out_cl (round 5): the result as NHWC fp32 rows (exactly one of out / (out_hi, out_lo) / out_cl).
static int halo_conv_launch(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                            int apply_relu, float* out, void* out_hi, void* out_lo, float* out_cl, int B, int C, int H, int W, int N,
                            const ff3d_scale_t* scale_host, ff3d_stream_t stream, int w_tiled = 0) {
  FF3D_REQUIRE(x_hi && x_lo && w_hi && w_lo && (out || (out_hi && out_lo) || out_cl), FF3D_ERR_NULL);
  FF3D_REQUIRE(!out_cl || (!out && !out_hi && !out_lo && ff3d_aligned16(out_cl)), FF3D_ERR_NULL);
  FF3D_REQUIRE(!scale_host || !scale_host->out_exp || scale_host->w_bound, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && C > 0 && C % HC_BK == 0 && H > 0 && W > 0 && N > 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(out || out_cl || N % 2 == 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(((long long)B * H * W + 1) * C * 2 < (1ll << 32) && ((long long)N + 1) * 9 * C * 2 < (1ll << 32),
               FF3D_ERR_BAD_SHAPE);
  const long long blocks = (long long)B * ((H + HC_Y - 1) / HC_Y) * ((W + HC_X - 1) / HC_X) * ((N + HC_BN - 1) / HC_BN);
  FF3D_REQUIRE(blocks < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  static bool configured[64] = {};              // per device: the > 64 KiB dynamic-LDS attribute is device state
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!configured[dev & 63]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_f16x3_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)HC_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_f16x3_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)HC_LDS_BYTES) != hipSuccess)
      return FF3D_ERR_LAUNCH;
    configured[dev & 63] = true;
  }
  HaloParams p{static_cast<const _Float16*>(x_hi), static_cast<const _Float16*>(x_lo),
               static_cast<const _Float16*>(w_hi), static_cast<const _Float16*>(w_lo), bias, out,
               static_cast<_Float16*>(out_hi), static_cast<_Float16*>(out_lo), B, C, H, W, N, apply_relu ? 1 : 0,
               (unsigned)((long long)B * H * W * C * 2), (unsigned)((long long)N * 9 * C * 2), ff3d_scale_from(scale_host)};
  p.out_cl = out_cl;
  p.w_tiled = w_tiled;
  ff3d_clear_error();
  static const bool no_tr_env = [] {                                      // tuning hook (as in splitmm.hip): FF3D_TR=none
    const char* e = getenv("FF3D_TR");
    return e && e[0] == 'n';
  }();
  const bool no_tr = no_tr_env && !out_cl;                                // (the channels-last form IS the transposed-tile epilogue)
#ifdef FF3D_BUILD_EXPERIMENTS
  static const bool m32 = [] {                                            // FF3D_HALO_M32=1: the 32x32x16 MFMA form (round 4)
    const char* e = getenv("FF3D_HALO_M32");
    return e && e[0] == '1';
  }();
  if (m32) {
    static bool configured_m[64] = {};
    if (!configured_m[dev & 63]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_m32_f16x3_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)HC_LDS_BYTES) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_m32_f16x3_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)HC_LDS_BYTES) != hipSuccess)
        return FF3D_ERR_LAUNCH;
      configured_m[dev & 63] = true;
    }
    if (!out && !no_tr)
      hipLaunchKernelGGL(conv3x3_halo_m32_f16x3_kernel<true>, dim3((unsigned)blocks), dim3(HC_T), HC_LDS_BYTES,
                         static_cast<hipStream_t>(stream), p);
    else
      hipLaunchKernelGGL(conv3x3_halo_m32_f16x3_kernel<false>, dim3((unsigned)blocks), dim3(HC_T), HC_LDS_BYTES,
                         static_cast<hipStream_t>(stream), p);
    return ff3d_launch_status();
  }
  static const bool wreg = [] {                                           // FF3D_HALO_WREG=1: weights from L1 / L2 into registers (round 4)
    const char* e = getenv("FF3D_HALO_WREG");
    return e && e[0] == '1';
  }();
  if (wreg) {
    constexpr size_t WR_LDS = (size_t)(2 * 2 * HC_ACT) * sizeof(_Float16);
    static bool configured_w[64] = {};
    if (!configured_w[dev & 63]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_wreg_f16x3_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)WR_LDS) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_wreg_f16x3_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)WR_LDS) != hipSuccess)
        return FF3D_ERR_LAUNCH;
      configured_w[dev & 63] = true;
    }
    if (!out && !no_tr)
      hipLaunchKernelGGL(conv3x3_halo_wreg_f16x3_kernel<true>, dim3((unsigned)blocks), dim3(HC_T), WR_LDS,
                         static_cast<hipStream_t>(stream), p);
    else
      hipLaunchKernelGGL(conv3x3_halo_wreg_f16x3_kernel<false>, dim3((unsigned)blocks), dim3(HC_T), WR_LDS,
                         static_cast<hipStream_t>(stream), p);
    return ff3d_launch_status();
  }
  static const int abl = [] {                                             // timing ablations: FF3D_HALO_ABLATE=bit mask
    const char* e = getenv("FF3D_HALO_ABLATE");
    return e ? atoi(e) : 0;
  }();
  if (abl) {
#define FF3D_ABL(n)                                                                                                         \
  case n: {                                                                                                                 \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_f16x3_kernel<false, n>),                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)HC_LDS_BYTES);                               \
    hipLaunchKernelGGL((conv3x3_halo_f16x3_kernel<false, n>), dim3((unsigned)blocks), dim3(HC_T), HC_LDS_BYTES,             \
                       static_cast<hipStream_t>(stream), p);                                                                \
    break;                                                                                                                  \
  }
    switch (abl) { FF3D_ABL(1) FF3D_ABL(2) FF3D_ABL(3) FF3D_ABL(4) FF3D_ABL(5) FF3D_ABL(6) FF3D_ABL(7) FF3D_ABL(8) FF3D_ABL(12) FF3D_ABL(13) FF3D_ABL(16) FF3D_ABL(32) FF3D_ABL(64) FF3D_ABL(96) FF3D_ABL(128) default: break; }
#undef FF3D_ABL
    return ff3d_launch_status();
  }
  static const bool halo8 = [] {                                          // FF3D_CONV_HALO8=1: the 8 x 64-pixel kernel (opt-in, see its header)
    const char* e = getenv("FF3D_CONV_HALO8");
    return e && e[0] == '1';
  }();
  if (halo8 && !(no_tr && !out)) {
    static bool configured8[64] = {};
    if (!configured8[dev & 63]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo8_f16x3_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)H8_LDS_BYTES) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo8_f16x3_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)H8_LDS_BYTES) != hipSuccess)
        return FF3D_ERR_LAUNCH;
      configured8[dev & 63] = true;
    }
    const long long blocks8 = (long long)B * ((H + H8_Y - 1) / H8_Y) * ((W + H8_X - 1) / H8_X) * ((N + HC_BN - 1) / HC_BN);
    if (out)
      hipLaunchKernelGGL(conv3x3_halo8_f16x3_kernel<false>, dim3((unsigned)blocks8), dim3(HC_T), H8_LDS_BYTES,
                         static_cast<hipStream_t>(stream), p);
    else
      hipLaunchKernelGGL(conv3x3_halo8_f16x3_kernel<true>, dim3((unsigned)blocks8), dim3(HC_T), H8_LDS_BYTES,
                         static_cast<hipStream_t>(stream), p);
    return ff3d_launch_status();
  }
  static const bool tile_order = [] {                                     // FF3D_MFMA_ORDER=tile: rounds 1-2 MFMA order (A/B runs)
    const char* e = getenv("FF3D_MFMA_ORDER");
    return e && e[0] == 't';
  }();
  if (tile_order) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_f16x3_kernel<true, 0, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)HC_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_f16x3_kernel<false, 0, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)HC_LDS_BYTES);
    if (!out && !no_tr)
      hipLaunchKernelGGL((conv3x3_halo_f16x3_kernel<true, 0, false>), dim3((unsigned)blocks), dim3(HC_T), HC_LDS_BYTES,
                         static_cast<hipStream_t>(stream), p);
    else
      hipLaunchKernelGGL((conv3x3_halo_f16x3_kernel<false, 0, false>), dim3((unsigned)blocks), dim3(HC_T), HC_LDS_BYTES,
                         static_cast<hipStream_t>(stream), p);
    return ff3d_launch_status();
  }
  static const int pipe = [] {    // FF3D_HALO_PIPE=1: the hand-scheduled form, 2: + de-phased DMA issue (opt-in: same results), 3-5: tuning ablations
    const char* e = getenv("FF3D_HALO_PIPE");
    return e ? atoi(e) : 0;
  }();
  if (pipe >= 1 && pipe <= 5) {
#define FF3D_PIPE(v)                                                                                                          \
  do {                                                                                                                        \
    static bool configured_p[64] = {};                                                                                        \
    if (!configured_p[dev & 63]) {                                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_pipe_f16x3_kernel<false, v>),                       \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)HP_LDS_BYTES) != hipSuccess ||                 \
          hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_pipe_f16x3_kernel<true, v>),                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)HP_LDS_BYTES) != hipSuccess)                   \
        return FF3D_ERR_LAUNCH;                                                                                               \
      configured_p[dev & 63] = true;                                                                                          \
    }                                                                                                                         \
    if (!out && !no_tr)                                                                                                       \
      hipLaunchKernelGGL((conv3x3_halo_pipe_f16x3_kernel<true, v>), dim3((unsigned)blocks), dim3(HC_T), HP_LDS_BYTES,         \
                         static_cast<hipStream_t>(stream), p);                                                                \
    else                                                                                                                      \
      hipLaunchKernelGGL((conv3x3_halo_pipe_f16x3_kernel<false, v>), dim3((unsigned)blocks), dim3(HC_T), HP_LDS_BYTES,        \
                         static_cast<hipStream_t>(stream), p);                                                                \
  } while (0)
    if (pipe == 1)
      FF3D_PIPE(0);
    else if (pipe == 2)
      FF3D_PIPE(1);
    else if (pipe == 3)
      FF3D_PIPE(2);
    else if (pipe == 4)
      FF3D_PIPE(4);
    else
      FF3D_PIPE(6);
#undef FF3D_PIPE
    return ff3d_launch_status();
  }
#endif  // FF3D_BUILD_EXPERIMENTS
  // tile geometry: 4 x 64 pixels, or 8 x 32 where that pads the map less (468 x 468: 512 columns against 480) - FF3D_HALO_GEO=0 | 1 forces
  static const int geo_force = [] {
    const char* e = getenv("FF3D_HALO_GEO");
    return e ? atoi(e) : -1;
  }();
  const long long pad0 = (long long)((H + 3) / 4 * 4) * ((W + 63) / 64 * 64), pad1 = (long long)((H + 7) / 8 * 8) * ((W + 31) / 32 * 32);
  // (the 8 x 32 form is ~4 % slower per padded pixel - 3.11 vs 2.93 ms at 180 x 180 - so it needs >= 5 % less padding:
  //  468 x 468 = 5.4 % -> 5.28 vs 5.35 ms, profiles/r03_z_halo_geometry_ab.txt)
  // (round 6: with two taps per barrier the 8 x 32 form costs the same per padded pixel - 2.95 - 2.98 ms for 184 x 192 against 2.89 for
  //  180 x 192 - so any real saving in padding takes it)
  const bool geo1 = geo_force >= 0 ? geo_force == 1 : pad1 * 100 < pad0 * 99;
  static const bool tap2 = [] {                                           // two taps per barrier in the 8 x 32 geometry (FF3D_HALO_TAP2=0: one)
    const char* e = getenv("FF3D_HALO_TAP2");
    return !(e && e[0] == '0');
  }();
  if (geo1 && tap2) {
    using G1 = HcGeo<1>;
    constexpr int LDS2 = (int)hc_tap2_lds_bytes<1>();
    static bool configured2[64] = {};
    if (!configured2[dev & 63]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_tap2_f16x3_kernel<false, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS2) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_tap2_f16x3_kernel<true, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS2) != hipSuccess)
        return FF3D_ERR_LAUNCH;
      configured2[dev & 63] = true;
    }
    const long long blocks1 = (long long)B * ((H + G1::TY - 1) / G1::TY) * ((W + G1::TX - 1) / G1::TX) * ((N + HC_BN - 1) / HC_BN);
    FF3D_REQUIRE(blocks1 < (1ll << 31), FF3D_ERR_BAD_SHAPE);
    if (!out && !no_tr)
      hipLaunchKernelGGL((conv3x3_halo_tap2_f16x3_kernel<true, 1>), dim3((unsigned)blocks1), dim3(HC_T), LDS2,
                         static_cast<hipStream_t>(stream), p);
    else
      hipLaunchKernelGGL((conv3x3_halo_tap2_f16x3_kernel<false, 1>), dim3((unsigned)blocks1), dim3(HC_T), LDS2,
                         static_cast<hipStream_t>(stream), p);
    return ff3d_launch_status();
  }
  if (geo1) {
    using G1 = HcGeo<1>;
    static bool configured1[64] = {};
    if (!configured1[dev & 63]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_f16x3_kernel<false, 0, true, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)G1::LDS_BYTES) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_f16x3_kernel<true, 0, true, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)G1::LDS_BYTES) != hipSuccess)
        return FF3D_ERR_LAUNCH;
      configured1[dev & 63] = true;
    }
    const long long blocks1 = (long long)B * ((H + G1::TY - 1) / G1::TY) * ((W + G1::TX - 1) / G1::TX) * ((N + HC_BN - 1) / HC_BN);
    FF3D_REQUIRE(blocks1 < (1ll << 31), FF3D_ERR_BAD_SHAPE);
    if (!out && !no_tr)
      hipLaunchKernelGGL((conv3x3_halo_f16x3_kernel<true, 0, true, 1>), dim3((unsigned)blocks1), dim3(HC_T), G1::LDS_BYTES,
                         static_cast<hipStream_t>(stream), p);
    else
      hipLaunchKernelGGL((conv3x3_halo_f16x3_kernel<false, 0, true, 1>), dim3((unsigned)blocks1), dim3(HC_T), G1::LDS_BYTES,
                         static_cast<hipStream_t>(stream), p);
    return ff3d_launch_status();
  }
  if (!out && !no_tr)
    hipLaunchKernelGGL(conv3x3_halo_f16x3_kernel<true>, dim3((unsigned)blocks), dim3(HC_T), HC_LDS_BYTES,
                       static_cast<hipStream_t>(stream), p);
  else
    hipLaunchKernelGGL(conv3x3_halo_f16x3_kernel<false>, dim3((unsigned)blocks), dim3(HC_T), HC_LDS_BYTES,
                       static_cast<hipStream_t>(stream), p);
  return ff3d_launch_status();
}

extern "C" int ff3d_conv3x3_halo_f16x3(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo,
                                       const float* bias, int apply_relu, float* out, void* out_hi, void* out_lo,
                                       int B, int C, int H, int W, int N, const ff3d_scale_t* scale_host,
                                       ff3d_stream_t stream) {
  return halo_conv_launch(x_hi, x_lo, w_hi, w_lo, bias, apply_relu, out, out_hi, out_lo, nullptr, B, C, H, W, N, scale_host, stream);
}

// The same convolution with the result as NHWC fp32 (B, H, W, N): what a channels-last consumer reads - the camera feature maps
// of `shared_conv_img` (necks/focal_encoder.py:143-147) go straight to the projection sampler's gather (EU:236-247) instead of
// through an NCHW tensor and a transposing pass.
extern "C" int ff3d_conv3x3_halo_f16x3_nhwc(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo,
                                            const float* bias, int apply_relu, float* out_nhwc, int B, int C, int H, int W,
                                            int N, const ff3d_scale_t* scale_host, ff3d_stream_t stream) {
  FF3D_REQUIRE(out_nhwc, FF3D_ERR_NULL);
  return halo_conv_launch(x_hi, x_lo, w_hi, w_lo, bias, apply_relu, nullptr, nullptr, nullptr, out_nhwc, B, C, H, W, N, scale_host,
                          stream);
}

// Round 5: the same two convolutions with the WEIGHT planes in K-step tiles, wt[tile][n][32] = w[n][tap][c0 .. c0 + 31], tile = tap *
// C / 32 + c0 / 32, N + 1 rows per tile (row N zeros): the 128 rows a block stages per (tap, channel chunk) step are one contiguous 8 KiB
// instead of 128 pieces of 64 bytes 9 C * 2 bytes apart - every 128-byte line of the weights is then used whole when it is fetched,
// not half now and half nine taps later.  A launch streams 10 GB of weights L2 -> LDS (profiles/r04_*); its own time does not move
// (3.15 - 3.18 vs 3.16 - 3.21 ms), the 32-frame step with two batches in flight does: 1307.1 / 1297.8 vs 1287.4 / 1286.7 frames/s on
// one box (profiles/r05_af_*).  Exactly one of out / (out_hi, out_lo) / out_nhwc is non-NULL.  Bit-identical results.
extern "C" int ff3d_conv3x3_halo_f16x3_tiled(const void* x_hi, const void* x_lo, const void* wt_hi, const void* wt_lo,
                                             const float* bias, int apply_relu, float* out, void* out_hi, void* out_lo,
                                             float* out_nhwc, int B, int C, int H, int W, int N,
                                             const ff3d_scale_t* scale_host, ff3d_stream_t stream) {
  FF3D_REQUIRE((out != nullptr) + (out_hi != nullptr || out_lo != nullptr) + (out_nhwc != nullptr) == 1, FF3D_ERR_NULL);
  return halo_conv_launch(x_hi, x_lo, wt_hi, wt_lo, bias, apply_relu, out, out_hi, out_lo, out_nhwc, B, C, H, W, N, scale_host, stream,
                          1);
}

// Round 6: ff3d_conv3x3_halo_f16x3_tiled over the caller's NCHW fp32 map x (B, C, H, W) - no fp32 -> pair conversion pass in front of the
// convolution (conv3x3_halo_nchw_f16x3_kernel above).  `hint` = the persistent exponent record of the call site (FF3D_SPLIT_HINT_INTS int32,
// zero-initialised, the record ff3d_split_f16 keeps): three launches - the conv with the guessed exponent, the one-wave check, the conv again
// (exits at once unless the check flagged it).  scale_host->a_exp is ignored (the kernels read hint[0]); w_exp / w_bound / out_exp as usual.
// 4 x 64 geometry only: FF3D_ERR_UNSUPPORTED where the launcher of the pair form would pick 8 x 32 (the caller then converts and calls that).
static int halo_conv_nchw_launch(const float* x, int32_t* hint, const void* wt_hi, const void* wt_lo, const float* bias, int apply_relu,
                                 float* out, void* out_hi, void* out_lo, int B, int C, int H, int W, int N, int nvar,
                                 const ff3d_scale_t* scale_host, ff3d_stream_t stream) {
  FF3D_REQUIRE(x && hint && wt_hi && wt_lo && ((out != nullptr) != (out_hi != nullptr && out_lo != nullptr)), FF3D_ERR_NULL);
  FF3D_REQUIRE(!scale_host || !scale_host->out_exp || scale_host->w_bound, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && C > 0 && C % HC_BK == 0 && H > 0 && W > 0 && N > 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(out || N % 2 == 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE((long long)C * H * W * 4 < (1ll << 31) && ((long long)N + 1) * 9 * C * 2 < (1ll << 32), FF3D_ERR_BAD_SHAPE);   // per-lane offsets < the buffer's 2^31 records
  FF3D_REQUIRE(!out_hi || ((long long)B * H * W + 1) * N * 2 < (1ll << 32), FF3D_ERR_BAD_SHAPE);
  const long long pad0 = (long long)((H + 3) / 4 * 4) * ((W + 63) / 64 * 64), pad1 = (long long)((H + 7) / 8 * 8) * ((W + 31) / 32 * 32);
  const bool geo1 = pad1 * 100 < pad0 * 99;                                             // (as halo_conv_launch: the 8 x 32 geometry pads less)
#ifndef FF3D_BUILD_EXPERIMENTS
  FF3D_REQUIRE(!geo1, FF3D_ERR_UNSUPPORTED);
#endif
  ff3d_scale_t sh = scale_host ? *scale_host : ff3d_scale_t{};
  sh.a_exp = hint;                                                                      // hint[0] = the exponent in use
  HaloParams p{nullptr, nullptr, static_cast<const _Float16*>(wt_hi), static_cast<const _Float16*>(wt_lo), bias, out,
               static_cast<_Float16*>(out_hi), static_cast<_Float16*>(out_lo), B, C, H, W, N, apply_relu ? 1 : 0, 0u,
               (unsigned)((long long)N * 9 * C * 2), ff3d_scale_from(&sh)};
  p.w_tiled = 1;
  p.x_f32 = x;
  p.x_hint = hint;
  int dev = 0;
  (void)hipGetDevice(&dev);
  ff3d_clear_error();
  hipStream_t s = static_cast<hipStream_t>(stream);
#define FF3D_NCHW_LAUNCH_G(TRV, GEOV, NV)                                                                                    \
  do {                                                                                                                       \
    using G = HcGeo<GEOV>;                                                                                                   \
    const long long blocks = (long long)B * ((H + G::TY - 1) / G::TY) * ((W + G::TX - 1) / G::TX) * ((N + HC_BN - 1) / HC_BN); \
    FF3D_REQUIRE(blocks < (1ll << 31), FF3D_ERR_BAD_SHAPE);                                                                  \
    static bool configured_n[64] = {};                                                                                       \
    if (!configured_n[dev & 63]) {                                                                                           \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_nchw_f16x3_kernel<TRV, GEOV, NV>),                 \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) != hipSuccess ||                \
          hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_nchw_redo_f16x3_kernel<TRV, GEOV, NV>),            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) != hipSuccess)                  \
        return FF3D_ERR_LAUNCH;                                                                                              \
      configured_n[dev & 63] = true;                                                                                         \
    }                                                                                                                        \
    p.x_redo = 0;                                                                                                            \
    hipLaunchKernelGGL((conv3x3_halo_nchw_f16x3_kernel<TRV, GEOV, NV>), dim3((unsigned)blocks), dim3(HC_T), G::LDS_BYTES, s, p); \
    hipLaunchKernelGGL(hc_nchw_verify_kernel, dim3(1), dim3(64), 0, s, hint);                                                \
    p.x_redo = 1;                                                                                                            \
    hipLaunchKernelGGL((conv3x3_halo_nchw_redo_f16x3_kernel<TRV, GEOV, NV>), dim3((unsigned)(blocks < 512 ? blocks : 512)),  \
                       dim3(HC_T), G::LDS_BYTES, s, p, (unsigned)blocks);                                                    \
  } while (0)
#define FF3D_NCHW_LAUNCH(TRV, NV) FF3D_NCHW_LAUNCH_G(TRV, 0, NV)
#ifdef FF3D_BUILD_EXPERIMENTS                     // (experiment: the 8 x 32 geometry, one tap per barrier, pair output - 468 x 468 maps)
  if (geo1) {
    FF3D_REQUIRE(!out && nvar == 9, FF3D_ERR_UNSUPPORTED);
    FF3D_NCHW_LAUNCH_G(true, 1, 9);
    return ff3d_launch_status();
  }
#endif
  switch (nvar) {
    case 9:                                       // the shipped form: pixel-fastest slots, requested / converted at the top of the step
      if (out) FF3D_NCHW_LAUNCH(false, 9); else FF3D_NCHW_LAUNCH(true, 9);
      break;
#ifdef FF3D_BUILD_EXPERIMENTS                     // (pair output only) the forms behind profiles/r06_nc_halo_nchw_source_ab.txt
    case 0: FF3D_REQUIRE(!out, FF3D_ERR_UNSUPPORTED); FF3D_NCHW_LAUNCH(true, 0); break;    // DMA slot order, behind the first MFMA pass
    case 1: FF3D_REQUIRE(!out, FF3D_ERR_UNSUPPORTED); FF3D_NCHW_LAUNCH(true, 1); break;    // DMA slot order, top of the step
    case 2: FF3D_REQUIRE(!out, FF3D_ERR_UNSUPPORTED); FF3D_NCHW_LAUNCH(true, 2); break;    // (WRONG results) no requests in the loop
    case 6: FF3D_REQUIRE(!out, FF3D_ERR_UNSUPPORTED); FF3D_NCHW_LAUNCH(true, 6); break;    // (WRONG results) neither requests nor conversion
    case 8: FF3D_REQUIRE(!out, FF3D_ERR_UNSUPPORTED); FF3D_NCHW_LAUNCH(true, 8); break;    // pixel-fastest, behind the first MFMA pass
    case 40: FF3D_REQUIRE(!out, FF3D_ERR_UNSUPPORTED); FF3D_NCHW_LAUNCH(true, 40); break;  // pixel-fastest, two register sets in flight
    case 73: FF3D_REQUIRE(!out, FF3D_ERR_UNSUPPORTED); FF3D_NCHW_LAUNCH(true, 73); break;  // 9 + non-temporal requests
    case 137: FF3D_REQUIRE(!out, FF3D_ERR_UNSUPPORTED); FF3D_NCHW_LAUNCH(true, 137); break; // 9 + requests ahead of the weight DMAs
#endif
    default: return FF3D_ERR_UNSUPPORTED;
  }
#undef FF3D_NCHW_LAUNCH_G
#undef FF3D_NCHW_LAUNCH
  return ff3d_launch_status();
}

extern "C" int ff3d_conv3x3_halo_f16x3_nchwsrc(const float* x, int32_t* hint, const void* wt_hi, const void* wt_lo, const float* bias,
                                               int apply_relu, float* out, void* out_hi, void* out_lo, int B, int C, int H, int W, int N,
                                               const ff3d_scale_t* scale_host, ff3d_stream_t stream) {
  return halo_conv_nchw_launch(x, hint, wt_hi, wt_lo, bias, apply_relu, out, out_hi, out_lo, B, C, H, W, N, 9, scale_host, stream);
}

#ifdef FF3D_BUILD_EXPERIMENTS
// (experiments library only; tools/experiments/exp_halo_nchw.py) the same with the tuning variant chosen by the caller
extern "C" int ff3d_exp_conv3x3_halo_nchwsrc(const float* x, int32_t* hint, const void* wt_hi, const void* wt_lo, const float* bias,
                                             int apply_relu, float* out, void* out_hi, void* out_lo, int B, int C, int H, int W, int N,
                                             int nvar, const ff3d_scale_t* scale_host, ff3d_stream_t stream) {
  return halo_conv_nchw_launch(x, hint, wt_hi, wt_lo, bias, apply_relu, out, out_hi, out_lo, B, C, H, W, N, nvar, scale_host, stream);
}
#endif  // FF3D_BUILD_EXPERIMENTS
