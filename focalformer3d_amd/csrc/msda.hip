// Multi-scale deformable attention forward for gfx950 (MI355X).
//
// Work decomposition (wave64): one "pair" = one (batch, query, head).  A pair is handled by LPG
// adjacent lanes, each owning one 16-byte slice of the head's Dh channels (4 x fp32 or 8 x bf16),
// so every bilinear corner is one fully-coalesced 16*LPG-byte row read of the channels-last value
// tensor and a 256-thread block covers 256/LPG pairs.  The per-pair sampling locations and
// attention weights (L*P*3 floats, shared by the LPG lanes) are staged once per block through LDS
// with coalesced loads; the fused variant also does the softmax and the ref + off/(W,H) prologue
// there.  Corner loads are branch-free (clamped address, zeroed weight) so the compiler can keep a
// whole level's 4*P loads in flight per lane - the kernel is a latency/bandwidth-bound gather, not
// GEMM-shaped, so no MFMA.  Blocks are remapped so each XCD's private L2 sees a contiguous range
// of frames.
//
// Algorithmic HBM bytes per call: B*Nq*heads*L*P*(4*Dh*sizeof(value) + 12) +
// B*Nq*heads*Dh*4.
#include <cstdlib>

#include "ff3d_common.h"

namespace {

struct MsdaParams {
  const void* value;
  const float* loc;      // plain: (npairs, LP, 2)      fused: raw offsets rows
  const float* attn_w;   // plain: (npairs, LP)         fused: raw logits rows
  const float* ref_pts;  // fused only: (B*Nq, 2)
  float* out;
  long long off_ld, logits_ld;
  long long cell_stride;  // elements between consecutive BEV cells of one frame (>= heads*Dh)
  int npairs, Nq, heads, Dh, P, LP;
  int Nv, L;
  LevelTable lv;                    // host-built level table (kernel argument), or - DEVLV kernels -
  const long long* shapes_dev;      // (L, 2) int64 (H_l, W_l) and
  const long long* starts_dev;      // (L) int64 level_start_index in device memory (the mmcv op ABI)
  long long out_ld;                 // SHARED instances: floats between consecutive (b, q) rows of `out`
  int hpg;                          // SHARED instances: heads per column group of `out` (each group: hpg * C channels + 32 tail columns)
};

// KIND 0: 4 x fp32 (16-byte loads)   KIND 1: 8 x bf16 (16-byte loads)   KIND 2: 1 x fp32 (Dh % 4 != 0)
template <int KIND>
struct Vec;
template <>
struct Vec<2> {
  static constexpr int N = 1;
  using load_t = float;
  using elem_t = float;
  __device__ static void fma(float* acc, const load_t& v, float w) { acc[0] = fmaf(w, v, acc[0]); }
};
template <>
struct Vec<0> {
  using elem_t = float;
  static constexpr int N = 4;
  using load_t = float4;
  __device__ static void fma(float* acc, const load_t& v, float w) {
    acc[0] = fmaf(w, v.x, acc[0]);
    acc[1] = fmaf(w, v.y, acc[1]);
    acc[2] = fmaf(w, v.z, acc[2]);
    acc[3] = fmaf(w, v.w, acc[3]);
  }
};
template <>
struct Vec<1> {
  using elem_t = unsigned short;
  static constexpr int N = 8;
  using load_t = uint4;
  __device__ static void fma(float* acc, const load_t& v, float w) {
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[2 * i] = fmaf(w, __uint_as_float(u[i] << 16), acc[2 * i]);
      acc[2 * i + 1] = fmaf(w, __uint_as_float(u[i] & 0xffff0000u), acc[2 * i + 1]);
    }
  }
};

// DEVLV: the level table comes from device memory (mmcv's spatial_shapes / level_start_index tensors) - staged into LDS
// next to the sampling locations, so the host never has to read those tensors (no sync, graph-capturable).
// PT: 0 = any number of points per level (loads issued next to their uses: the compiler serialises most of them, each waited for
// with vmcnt(0) - latency is hidden by occupancy alone, 63 VGPRs = 8 waves per SIMD); 4 (round 4, P == 4 = every shipped config):
// the 16 corner loads of a level's four points are issued back to back before the first of them is consumed.
// SHARED (round 6, the opt-in "gather first" value mode: ff3d_msda_gather_rows): every head gathers the WHOLE C-wide row of the
// un-projected value (B, Nv, C) at its own sampling locations - a pair = (b, q, head) still, Dh = C lanes-worth of channels, the
// value base does not depend on the head - into out[(b, q)][head * C + c]; the pair's sum of valid corner weights (what the
// projection's bias has to be multiplied with: out-of-map corners contribute neither value nor bias) goes to out[(b, q)][heads * C +
// head], head 0 also zeroes the padding columns up to heads * C + 32.  value_proj is applied AFTER the gather (one block-diagonal GEMM).
template <int LPG, bool FUSED, int KIND, bool DEVLV = false, int PT = 0, bool SHARED = false>
__global__ __launch_bounds__(256) void msda_fwd_kernel(MsdaParams p) {
  constexpr int PPB = 256 / LPG;  // pairs per block
  using V = Vec<KIND>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_lv[3 * FF3D_MAX_LEVELS];
  if (DEVLV && threadIdx.x < (unsigned)p.L) {
    s_lv[threadIdx.x] = (int)p.shapes_dev[2 * threadIdx.x];
    s_lv[FF3D_MAX_LEVELS + threadIdx.x] = (int)p.shapes_dev[2 * threadIdx.x + 1];
    s_lv[2 * FF3D_MAX_LEVELS + threadIdx.x] = (int)p.starts_dev[threadIdx.x];
  }
  static_assert(!(DEVLV && FUSED), "the fused prologue reads the level sizes before the staging barrier");
  float* s_loc = smem;                  // [PPB][LP][2]
  float* s_w = smem + PPB * p.LP * 2;   // [PPB][LP]

  const unsigned bid = ff3d_xcd_remap(blockIdx.x, gridDim.x);
  const int pair0 = bid * PPB;
  const int npair_blk = min(PPB, p.npairs - pair0);
  const int LP = p.LP;
  const int tid = threadIdx.x;

  // ---- stage sampling locations and weights for the block's pairs through LDS
  if (!FUSED) {
    const float* gl = p.loc + (long long)pair0 * LP * 2;
    const float* gw = p.attn_w + (long long)pair0 * LP;
    const int nl = npair_blk * LP * 2, nw = npair_blk * LP;
    // pair0*LP*2 floats is a multiple of 4 floats whenever PPB*LP*2 % 4 == 0 (always: PPB >= 4)
    for (int i = tid * 4; i < nl; i += 256 * 4) {
      if (i + 3 < nl) {
        *reinterpret_cast<float4*>(s_loc + i) = *reinterpret_cast<const float4*>(gl + i);
      } else {
        for (int j = i; j < nl; ++j) s_loc[j] = gl[j];
      }
    }
    for (int i = tid; i < nw; i += 256) s_w[i] = gw[i];
  } else {
    const int per = LP * 2;
    for (int e = tid; e < npair_blk * per; e += 256) {
      const int pl = e / per, r = e - pl * per;
      const int pg = pair0 + pl;
      const int row = pg / p.heads, h = pg - row * p.heads;
      const float o = p.loc[(long long)row * p.off_ld + h * per + r];
      const int pt = r >> 1, l = pt / p.P;
      const float norm = (r & 1) ? (float)p.lv.H[l] : (float)p.lv.W[l];
      s_loc[e] = p.ref_pts[row * 2 + (r & 1)] + o / norm;
    }
    for (int e = tid; e < npair_blk * LP; e += 256) {
      const int pl = e / LP, r = e - pl * LP;
      const int pg = pair0 + pl;
      const int row = pg / p.heads, h = pg - row * p.heads;
      s_w[e] = p.attn_w[(long long)row * p.logits_ld + h * LP + r];
    }
  }
  __syncthreads();

  const int pl = tid / LPG, sub = tid - pl * LPG;
  if (pl >= npair_blk) return;
  const int pair = pair0 + pl;
  const int row = pair / p.heads, h = pair - row * p.heads;  // row = b*Nq + q
  const int b = row / p.Nq;

  const float* ploc = s_loc + pl * LP * 2;
  const float* pw = s_w + pl * LP;
  float wmax = 0.f, winv = 1.f;
  if (FUSED) {  // softmax over the L*P logits of this pair (redundantly per lane, LP <= 64)
    wmax = pw[0];
    for (int i = 1; i < LP; ++i) wmax = fmaxf(wmax, pw[i]);
    float s = 0.f;
    for (int i = 0; i < LP; ++i) s += expf(pw[i] - wmax);
    winv = 1.f / s;
  }

  using elem_t = typename V::elem_t;
  const long long cell_stride = p.cell_stride;
  const elem_t* vbase =
      reinterpret_cast<const elem_t*>(p.value) + (long long)b * p.Nv * cell_stride + (SHARED ? 0 : h * p.Dh) + sub * V::N;
  float wsum = 0.f;

  float acc[V::N];
#pragma unroll
  for (int i = 0; i < V::N; ++i) acc[i] = 0.f;

  for (int l = 0; l < p.L; ++l) {
    const int Hl = DEVLV ? s_lv[l] : p.lv.H[l], Wl = DEVLV ? s_lv[FF3D_MAX_LEVELS + l] : p.lv.W[l];
    const elem_t* vl = vbase + (long long)(DEVLV ? s_lv[2 * FF3D_MAX_LEVELS + l] : p.lv.start[l]) * cell_stride;
    const float fH = (float)Hl, fW = (float)Wl;
    auto point = [&](int pt, float (&cw)[4], const elem_t* (&cp)[4]) {
      const int k = l * p.P + pt;
      const float x = ploc[2 * k], y = ploc[2 * k + 1];
      float aw = pw[k];
      if (FUSED) aw = expf(aw - wmax) * winv;
      // pixel coordinates, align_corners=False; clamp so the int conversion is always defined
      const float w_im = fminf(fmaxf(x * fW - 0.5f, -2.f), fW + 1.f);
      const float h_im = fminf(fmaxf(y * fH - 0.5f, -2.f), fH + 1.f);
      const float h_lo = floorf(h_im), w_lo = floorf(w_im);
      const float lh = h_im - h_lo, lw = w_im - w_lo, hh = 1.f - lh, hw = 1.f - lw;
      const int y0 = (int)h_lo, x0 = (int)w_lo, y1 = y0 + 1, x1 = x0 + 1;
      const bool vy0 = (unsigned)y0 < (unsigned)Hl, vy1 = (unsigned)y1 < (unsigned)Hl;
      const bool vx0 = (unsigned)x0 < (unsigned)Wl, vx1 = (unsigned)x1 < (unsigned)Wl;
      cw[0] = (vy0 && vx0) ? hh * hw * aw : 0.f;
      cw[1] = (vy0 && vx1) ? hh * lw * aw : 0.f;
      cw[2] = (vy1 && vx0) ? lh * hw * aw : 0.f;
      cw[3] = (vy1 && vx1) ? lh * lw * aw : 0.f;
      if (SHARED) wsum += (cw[0] + cw[1]) + (cw[2] + cw[3]);
      const int cy0 = min(max(y0, 0), Hl - 1), cy1 = min(max(y1, 0), Hl - 1);
      const int cx0 = min(max(x0, 0), Wl - 1), cx1 = min(max(x1, 0), Wl - 1);
      cp[0] = vl + (long long)(cy0 * Wl + cx0) * cell_stride;
      cp[1] = vl + (long long)(cy0 * Wl + cx1) * cell_stride;
      cp[2] = vl + (long long)(cy1 * Wl + cx0) * cell_stride;
      cp[3] = vl + (long long)(cy1 * Wl + cx1) * cell_stride;
    };
    if constexpr (PT == 4) {
      float cw[4][4];
      typename V::load_t cv[4][4];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        const elem_t* cp[4];
        point(pt, cw[pt], cp);
#pragma unroll
        for (int c = 0; c < 4; ++c) cv[pt][c] = *reinterpret_cast<const typename V::load_t*>(cp[c]);
      }
      __builtin_amdgcn_sched_barrier(0);           // all 16 loads are in flight before the first one is consumed
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int c = 0; c < 4; ++c) V::fma(acc, cv[pt][c], cw[pt][c]);
    } else {
#pragma unroll 4
      for (int pt = 0; pt < p.P; ++pt) {
        float cw[4];
        const elem_t* cp[4];
        point(pt, cw, cp);
        const typename V::load_t v00 = *reinterpret_cast<const typename V::load_t*>(cp[0]);
        const typename V::load_t v01 = *reinterpret_cast<const typename V::load_t*>(cp[1]);
        const typename V::load_t v10 = *reinterpret_cast<const typename V::load_t*>(cp[2]);
        const typename V::load_t v11 = *reinterpret_cast<const typename V::load_t*>(cp[3]);
        V::fma(acc, v00, cw[0]);
        V::fma(acc, v01, cw[1]);
        V::fma(acc, v10, cw[2]);
        V::fma(acc, v11, cw[3]);
      }
    }
  }

  float* o = p.out + (long long)pair * p.Dh + sub * V::N;
  if (SHARED) {
    // column group g = h / hpg: [hpg heads x C channels | hpg sums of valid weights | zeros up to 32]
    const int g = h / p.hpg, hl = h - g * p.hpg;
    float* grp = p.out + (long long)row * p.out_ld + (long long)g * (p.hpg * p.Dh + 32);
    o = grp + (long long)hl * p.Dh + sub * V::N;
    float* tail = grp + (long long)p.hpg * p.Dh;
    if (sub == 0) tail[hl] = wsum;
    if (hl == 0)
      for (int c = sub; c < 32; c += LPG)
        if (c >= p.hpg) tail[c] = 0.f;
  }
  if constexpr (V::N == 1) {
    o[0] = acc[0];
  } else {
    *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if constexpr (V::N == 8) *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

template <bool FUSED, int KIND, bool DEVLV = false>
int launch_lpg(int lpg, const MsdaParams& p, hipStream_t s) {
  const int ppb = 256 / lpg;
  const size_t smem = (size_t)ppb * p.LP * 3 * sizeof(float);
  if (smem > 64 * 1024) return FF3D_ERR_UNSUPPORTED;
  const unsigned grid = (p.npairs + ppb - 1) / ppb;
#define FF3D_MSDA_CASE(N)                                                                    \
  case N:                                                                                    \
    if (p.P == 4 && pt4)                                                                     \
      hipLaunchKernelGGL((msda_fwd_kernel<N, FUSED, KIND, DEVLV, 4>), dim3(grid), dim3(256), smem, s, p); \
    else                                                                                     \
      hipLaunchKernelGGL((msda_fwd_kernel<N, FUSED, KIND, DEVLV>), dim3(grid), dim3(256), smem, s, p); \
    break;
  // the batched-load instance pays while the gather is latency-bound (4 frames: 21 vs 26-28 us) and not once it is bandwidth-
  // bound (32 frames: 125-135 vs 123-124 us, 3 waves per SIMD instead of 8): used up to 64 K (query, head) pairs = 13 frames.
  // A/B hook: FF3D_MSDA_PT4 = 0 never | 1 always (profiles/r04_m_msda_batched_loads_ab.txt)
  static const int pt4_force = [] {
    const char* e = getenv("FF3D_MSDA_PT4");
    return e ? (e[0] == '0' ? 0 : 1) : -1;
  }();
  const bool pt4 = pt4_force >= 0 ? pt4_force == 1 : p.npairs <= 65536;
  ff3d_clear_error();
  switch (lpg) {
    FF3D_MSDA_CASE(1)
    FF3D_MSDA_CASE(2)
    FF3D_MSDA_CASE(4)
    FF3D_MSDA_CASE(8)
    FF3D_MSDA_CASE(16)
    FF3D_MSDA_CASE(32)
    FF3D_MSDA_CASE(64)
    default:
      return FF3D_ERR_BAD_SHAPE;
  }
#undef FF3D_MSDA_CASE
  return ff3d_launch_status();
}

int msda_dispatch(bool fused, const void* value, int value_dtype, long long value_ld, const float* a0,
                  const float* a1, const float* ref_pts, long long off_ld, long long logits_ld, float* out, int B, int Nv, int Nq,
                  int heads, int Dh, int L, int P, const int32_t* level_hw_host, ff3d_stream_t stream,
                  const int64_t* shapes_dev = nullptr, const int64_t* starts_dev = nullptr) {
  FF3D_REQUIRE(value && a0 && a1 && out && (!fused || ref_pts), FF3D_ERR_NULL);
  const bool devlv = shapes_dev != nullptr;
  FF3D_REQUIRE(!devlv || (starts_dev && !fused), FF3D_ERR_NULL);
  FF3D_REQUIRE(value_dtype == FF3D_F32 || value_dtype == FF3D_BF16, FF3D_ERR_BAD_DTYPE);
  FF3D_REQUIRE(B > 0 && Nv > 0 && Nq > 0 && heads > 0 && Dh > 0 && L > 0 && P > 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(L <= FF3D_MAX_LEVELS && L * P <= 64, FF3D_ERR_BAD_SHAPE);
  int kind = value_dtype == FF3D_BF16 ? 1 : 0;
  int vec = value_dtype == FF3D_BF16 ? 8 : 4;
  if (value_dtype == FF3D_F32 && Dh % 4 != 0) {  // tiny heads: scalar loads, one lane per channel
    kind = 2;
    vec = 1;
  }
  FF3D_REQUIRE(Dh % vec == 0, FF3D_ERR_BAD_SHAPE);
  const int lpg = Dh / vec;
  FF3D_REQUIRE(lpg <= 64 && (lpg & (lpg - 1)) == 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE((long long)B * Nq * heads < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(value) && ff3d_aligned16(out), FF3D_ERR_ALIGNMENT);
  if (!fused) FF3D_REQUIRE(ff3d_aligned16(a0), FF3D_ERR_ALIGNMENT);
  MsdaParams p;
  if (devlv) {
    for (int l = 0; l < FF3D_MAX_LEVELS; ++l) p.lv.H[l] = p.lv.W[l] = p.lv.start[l] = 0;
    p.lv.L = L, p.lv.Nv = Nv;
  } else {
    FF3D_REQUIRE(ff3d_make_levels(level_hw_host, L, &p.lv) && p.lv.Nv == Nv, FF3D_ERR_BAD_SHAPE);
  }
  p.Nv = Nv;
  p.L = L;
  p.shapes_dev = reinterpret_cast<const long long*>(shapes_dev);
  p.starts_dev = reinterpret_cast<const long long*>(starts_dev);
  p.value = value;
  p.loc = a0;
  p.attn_w = a1;
  p.ref_pts = ref_pts;
  p.out = out;
  p.off_ld = off_ld;
  p.logits_ld = logits_ld;
  p.out_ld = 0, p.hpg = 1;
  p.cell_stride = value_ld ? value_ld : (long long)heads * Dh;
  FF3D_REQUIRE(p.cell_stride >= (long long)heads * Dh && p.cell_stride % vec == 0, FF3D_ERR_BAD_SHAPE);
  p.npairs = B * Nq * heads;
  p.Nq = Nq;
  p.heads = heads;
  p.Dh = Dh;
  p.P = P;
  p.LP = L * P;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (devlv)
    return kind == 1 ? launch_lpg<false, 1, true>(lpg, p, s) : kind == 2 ? launch_lpg<false, 2, true>(lpg, p, s)
                                                                          : launch_lpg<false, 0, true>(lpg, p, s);
  if (fused) {
    return kind == 1 ? launch_lpg<true, 1>(lpg, p, s) : kind == 2 ? launch_lpg<true, 2>(lpg, p, s) : launch_lpg<true, 0>(lpg, p, s);
  }
  return kind == 1 ? launch_lpg<false, 1>(lpg, p, s) : kind == 2 ? launch_lpg<false, 2>(lpg, p, s) : launch_lpg<false, 0>(lpg, p, s);
}

}  // namespace

extern "C" int ff3d_msda_fwd(const void* value, int value_dtype, const float* loc, const float* attn_w, float* out,
                             int B, int Nv, int Nq, int heads, int Dh, int L, int P, const int32_t* level_hw_host,
                             ff3d_stream_t stream) {
  return msda_dispatch(false, value, value_dtype, 0, loc, attn_w, nullptr, 0, 0, out, B, Nv, Nq, heads, Dh, L, P,
                       level_hw_host, stream);
}

extern "C" int ff3d_msda_fused_fwd(const void* value, int value_dtype, int64_t value_ld, const float* ref_pts, const float* off,
                                   int64_t off_ld, const float* logits, int64_t logits_ld, float* out, int B, int Nv,
                                   int Nq, int heads, int Dh, int L, int P, const int32_t* level_hw_host,
                                   ff3d_stream_t stream) {
  FF3D_REQUIRE(off_ld >= (int64_t)heads * L * P * 2 && logits_ld >= (int64_t)heads * L * P, FF3D_ERR_BAD_SHAPE);
  return msda_dispatch(true, value, value_dtype, value_ld, off, logits, ref_pts, off_ld, logits_ld, out, B, Nv, Nq, heads, Dh,
                       L, P, level_hw_host, stream);
}

// Round 6, opt-in value mode "gather first" (value_proj is linear: sum_k w_k (W v_k + b) = W (sum_k w_k v_k) + b sum_k w_k): the gather of
// the UN-projected C-wide rows, per (query, head), with the fused prologue of ff3d_msda_fused_fwd.  out rows of out_ld >= groups * (heads / groups * C
// + 32) floats, `groups` column groups of heads / groups heads each: [the group's heads' C channels each | their sums of valid weights |
// zeros up to + 32] - one group per diagonal block of the projection that follows (groups = 2: two K = 4 C + 32 products in one dual
// launch instead of one K = 8 C + 32 product with half of its weight zeros).
extern "C" int ff3d_msda_gather_rows(const float* value, const float* ref_pts, const float* off, int64_t off_ld, const float* logits,
                                     int64_t logits_ld, float* out, int64_t out_ld, int groups, int B, int Nv, int Nq, int heads,
                                     int C, int L, int P, const int32_t* level_hw_host, ff3d_stream_t stream) {
  FF3D_REQUIRE(value && ref_pts && off && logits && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && Nv > 0 && Nq > 0 && heads > 0 && heads <= 32 && C > 0 && L > 0 && P > 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(L <= FF3D_MAX_LEVELS && L * P <= 64 && (C == 256 || C == 128 || C == 64), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(groups >= 1 && heads % groups == 0 && heads / groups <= 32, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(off_ld >= (int64_t)heads * L * P * 2 && logits_ld >= (int64_t)heads * L * P &&
                   out_ld >= (int64_t)heads * C + 32 * groups && out_ld % 4 == 0,
               FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE((long long)B * Nq * heads < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(value) && ff3d_aligned16(out), FF3D_ERR_ALIGNMENT);
  MsdaParams p;
  FF3D_REQUIRE(ff3d_make_levels(level_hw_host, L, &p.lv) && p.lv.Nv == Nv, FF3D_ERR_BAD_SHAPE);
  p.Nv = Nv, p.L = L, p.shapes_dev = nullptr, p.starts_dev = nullptr;
  p.value = value, p.loc = off, p.attn_w = logits, p.ref_pts = ref_pts, p.out = out;
  p.off_ld = off_ld, p.logits_ld = logits_ld, p.out_ld = out_ld, p.hpg = heads / groups;
  p.cell_stride = C;
  p.npairs = B * Nq * heads, p.Nq = Nq, p.heads = heads, p.Dh = C, p.P = P, p.LP = L * P;
  const int lpg = C / 4, ppb = 256 / lpg;
  const size_t smem = (size_t)ppb * p.LP * 3 * sizeof(float);
  const unsigned grid = (p.npairs + ppb - 1) / ppb;
  hipStream_t s = static_cast<hipStream_t>(stream);
  ff3d_clear_error();
  if (C == 256)
    hipLaunchKernelGGL((msda_fwd_kernel<64, true, 0, false, 0, true>), dim3(grid), dim3(256), smem, s, p);
  else if (C == 128)
    hipLaunchKernelGGL((msda_fwd_kernel<32, true, 0, false, 0, true>), dim3(grid), dim3(256), smem, s, p);
  else
    hipLaunchKernelGGL((msda_fwd_kernel<16, true, 0, false, 0, true>), dim3(grid), dim3(256), smem, s, p);
  return ff3d_launch_status();
}

// The mmcv op ABI itself: spatial_shapes / level_start_index stay DEVICE int64 tensors, exactly what
// ext_module.ms_deform_attn_forward receives - no host copy of the level table, no synchronisation.
extern "C" int ff3d_msda_fwd_dev(const void* value, int value_dtype, const int64_t* spatial_shapes_dev,
                                 const int64_t* level_start_index_dev, const float* loc, const float* attn_w, float* out,
                                 int B, int Nv, int Nq, int heads, int Dh, int L, int P, ff3d_stream_t stream) {
  FF3D_REQUIRE(spatial_shapes_dev && level_start_index_dev, FF3D_ERR_NULL);
  return msda_dispatch(false, value, value_dtype, 0, loc, attn_w, nullptr, 0, 0, out, B, Nv, Nq, heads, Dh, L, P, nullptr,
                       stream, spatial_shapes_dev, level_start_index_dev);
}
