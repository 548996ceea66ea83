// Hard-Instance-Probing stage kernels for gfx950: fused sigmoid*mask + 3x3 local-max NMS +
// score histogram; deterministic per-sample top-k (histogram threshold -> candidate compaction ->
// LDS bitonic sort); fused query gathers + positive-mask update.
//
// All of this is HBM/L2-bound integer/float bookkeeping on (B, K, H, W) fp32 maps (1.3 MB per
// frame at K=10, 180x180) - no GEMM shape anywhere, so no MFMA.
#include <cstdlib>

#include "ff3d_common.h"

namespace {

// Monotone (non-decreasing in s) linear bin of a score in [0, 1]; MUST be the same expression in
// the histogram producer and the top-k consumer.
__device__ __forceinline__ int score_bin(float s) {
  const int b = __float2int_rz(s * (float)FF3D_HIST_BINS);
  return min(max(b, 0), FF3D_HIST_BINS - 1);
}

__device__ __forceinline__ float sigmoidf_exact(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------------------------------------
// NMS: one block = an 8-row strip of one (b, class) plane in 32x8 tiles, halo tile in LDS.
constexpr int TX = 32, TY = 8;

__global__ __launch_bounds__(256) void heatmap_nms_kernel(const float* __restrict__ logits,
                                                          const float* __restrict__ logits_b,
                                                          const float* __restrict__ mask_in,
                                                          float* __restrict__ mask_next, float* __restrict__ heat,
                                                          uint32_t* __restrict__ hist, int K, int H, int W,
                                                          int nms_kernel, uint32_t small_bits) {
  __shared__ float tile[TY + 2][TX + 2 + 1];
  __shared__ uint32_t lhist[FF3D_HIST_BINS];
  // one block = an 8-row strip of one (b, class) plane, walked in 32-column tiles: the 4096-bin LDS histogram is cleared
  // and flushed once per strip (per 256-cell tile it cost ten times the NMS work itself)
  const int tiles_x = (W + TX - 1) / TX;
  const int ty0 = blockIdx.x * TY;
  const int cls = blockIdx.y, b = blockIdx.z;
  const long long plane = ((long long)b * K + cls) * H * W;
  const int tid = threadIdx.x;

  for (int i = tid; i < FF3D_HIST_BINS; i += 256) lhist[i] = 0;

  for (int txi = 0; txi < tiles_x; ++txi) {
    const int tx0 = txi * TX;
    __syncthreads();                             // previous tile's reads done (and, first time, lhist cleared)
    // halo tile of h = sigmoid(logit) * mask  (or the two-heatmap mean, FD:549)
    for (int i = tid; i < (TY + 2) * (TX + 2); i += 256) {
      const int ly = i / (TX + 2), lx = i - ly * (TX + 2);
      const int y = ty0 + ly - 1, x = tx0 + lx - 1;
      float h = 0.f;
      if (y >= 0 && y < H && x >= 0 && x < W) {
        const long long o = plane + (long long)y * W + x;
        h = sigmoidf_exact(logits[o]);
        if (logits_b) h = (h + sigmoidf_exact(logits_b[o])) / 2.f;
        if (mask_in) h = h * mask_in[o];
      }
      tile[ly][lx] = h;
    }
    __syncthreads();

    const int lx = tid % TX, ly = tid / TX;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x < W && y < H) {
      const float h = tile[ly + 1][lx + 1];
      float r = h;
      const bool small = (small_bits >> cls) & 1u;
      if (nms_kernel == 3 && !small) {
        // FD:673-676: local_max is 0 on the border ring, the valid 3x3 max inside
        if (x == 0 || y == 0 || x == W - 1 || y == H - 1) {
          r = (h == 0.f) ? h : 0.f;
        } else {
          float m = h;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) m = fmaxf(m, tile[ly + dy][lx + dx]);
          r = (h == m) ? h : 0.f;
        }
      }
      const long long o = plane + (long long)y * W + x;
      heat[o] = r;
      if (mask_next) mask_next[o] = mask_in ? mask_in[o] : 1.f;
      if (r > 0.f) atomicAdd(&lhist[score_bin(r)], 1u);
    }
  }
  __syncthreads();
  uint32_t* gh = hist + (long long)b * FF3D_HIST_BINS;
  for (int i = tid; i < FF3D_HIST_BINS; i += 256) {
    const uint32_t c = lhist[i];
    if (c) atomicAdd(&gh[i], c);
  }
}

// Round 4: the same arithmetic on full-width strips with 16-byte accesses (W % 4 == 0 and 16-byte aligned planes; the kernel
// above serves every other shape).  The 32 x 8-cell tiles above re-read 1.33 x the plane (10 x 34 halo per 8 x 32 tile), load and
// store 4 bytes per lane with an integer division per halo cell, pad the 180-wide rows to 192 and synchronise twice per
// tile: 88 us for the 166 MB of a 32-frame stage = 0.23 of the HBM peak.  Here a block owns SY rows x up to CW columns:
//   phase 1  every row of the strip (+ one halo row above and below) as float4: sigmoid(logit) (* mask) -> LDS; the interior
//            rows' mask clone is written from the registers that hold it;
//   phase 2  one float4 of outputs per thread: 3 x (float4 + 2 neighbours) from LDS, 3 x 3 max, exact == test, float4 store,
//            LDS histogram.
// Halo rows cost (SY + 2) / SY = 1.17 x reads at SY = 12 (180 = 15 x 12, 468 = 39 x 12: no padded strip either).
constexpr int NV_SY = 12, NV_CW = 256, NV_TW = NV_CW + 8;          // tile row: [4 pad | CW cells | 4 pad], 16-byte aligned groups

__global__ __launch_bounds__(256) void heatmap_nms_wide_kernel(const float* __restrict__ logits,
                                                               const float* __restrict__ logits_b,
                                                               const float* __restrict__ mask_in,
                                                               float* __restrict__ mask_next, float* __restrict__ heat,
                                                               uint32_t* __restrict__ hist, int K, int H, int W,
                                                               int nms_kernel, uint32_t small_bits) {
  __shared__ __attribute__((aligned(16))) float tile[NV_SY + 2][NV_TW];
  __shared__ uint32_t lhist[FF3D_HIST_BINS];
  const int ty0 = blockIdx.x * NV_SY;
  const int cls = blockIdx.y, b = blockIdx.z;
  const long long plane = ((long long)b * K + cls) * H * W;
  const int tid = threadIdx.x;
  const bool plain = nms_kernel != 3 || ((small_bits >> cls) & 1u);     // kernel-1 classes: every cell is its own maximum
  // LDS-privatised histogram, flushed with one coalesced atomic per non-empty bin (lane = bin).  Tried in round 4 and slower
  // (profiles/r04_g_*, r04_h_*): survivors straight into the frame's global histogram (245 us against 65 at 32 frames), and a
  // lane = 4-bins flush with a (column, row-group) thread mapping (144 us) - the global atomics are the expensive part of this
  // kernel, and they are cheapest when consecutive lanes hit consecutive counters.
  for (int i = tid; i < FF3D_HIST_BINS; i += 256) lhist[i] = 0;
  const int rows = min(NV_SY, H - ty0);

  for (int x0 = 0; x0 < W; x0 += NV_CW) {
    const int cw = min(NV_CW, W - x0), cw4 = cw >> 2;                    // (W % 4 == 0)
    __syncthreads();                                                     // previous chunk's reads done / lhist cleared
    // ---- phase 1: h = sigmoid(logit) (* mask) for rows ty0 - 1 .. ty0 + rows, columns x0 .. x0 + cw - 1 (+ the two halo columns)
    for (int i = tid; i < (rows + 2) * cw4; i += 256) {
      const int ry = i / cw4, c4 = i - ry * cw4;
      const int y = ty0 + ry - 1;
      float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
      if (y >= 0 && y < H) {
        const long long o = plane + (long long)y * W + x0 + 4 * c4;
        const float4 l = *reinterpret_cast<const float4*>(logits + o);
        h = make_float4(sigmoidf_exact(l.x), sigmoidf_exact(l.y), sigmoidf_exact(l.z), sigmoidf_exact(l.w));
        if (logits_b) {
          const float4 l2 = *reinterpret_cast<const float4*>(logits_b + o);
          h.x = (h.x + sigmoidf_exact(l2.x)) / 2.f, h.y = (h.y + sigmoidf_exact(l2.y)) / 2.f;
          h.z = (h.z + sigmoidf_exact(l2.z)) / 2.f, h.w = (h.w + sigmoidf_exact(l2.w)) / 2.f;
        }
        float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
        if (mask_in) {
          m = *reinterpret_cast<const float4*>(mask_in + o);
          h.x *= m.x, h.y *= m.y, h.z *= m.z, h.w *= m.w;
        }
        if (mask_next && ry >= 1 && ry <= rows) *reinterpret_cast<float4*>(mask_next + o) = m;
      }
      *reinterpret_cast<float4*>(&tile[ry][4 + 4 * c4]) = h;
    }
    if (!plain)
      for (int i = tid; i < (rows + 2) * 2; i += 256) {                  // halo columns x0 - 1 and x0 + cw
        const int ry = i >> 1, side = i & 1;
        const int y = ty0 + ry - 1, x = side ? x0 + cw : x0 - 1;
        float h = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
          const long long o = plane + (long long)y * W + x;
          h = sigmoidf_exact(logits[o]);
          if (logits_b) h = (h + sigmoidf_exact(logits_b[o])) / 2.f;
          if (mask_in) h = h * mask_in[o];
        }
        tile[ry][side ? 4 + cw : 3] = h;
      }
    __syncthreads();
    // ---- phase 2: one float4 of outputs per thread
    for (int i = tid; i < rows * cw4; i += 256) {
      const int ly = i / cw4, c4 = i - ly * cw4;
      const int y = ty0 + ly, xb = x0 + 4 * c4;
      const float4 c = *reinterpret_cast<const float4*>(&tile[ly + 1][4 + 4 * c4]);
      float hv[4] = {c.x, c.y, c.z, c.w}, r[4] = {c.x, c.y, c.z, c.w};
      if (!plain) {
        float colmax[6];                                                 // column maxima over the three rows, columns xb - 1 .. xb + 4
#pragma unroll
        for (int q = 0; q < 6; ++q) colmax[q] = hv[0];
        {
          const float4 u = *reinterpret_cast<const float4*>(&tile[ly][4 + 4 * c4]);
          const float4 d = *reinterpret_cast<const float4*>(&tile[ly + 2][4 + 4 * c4]);
          colmax[1] = fmaxf(fmaxf(u.x, c.x), d.x), colmax[2] = fmaxf(fmaxf(u.y, c.y), d.y);
          colmax[3] = fmaxf(fmaxf(u.z, c.z), d.z), colmax[4] = fmaxf(fmaxf(u.w, c.w), d.w);
          colmax[0] = fmaxf(fmaxf(tile[ly][3 + 4 * c4], tile[ly + 1][3 + 4 * c4]), tile[ly + 2][3 + 4 * c4]);
          colmax[5] = fmaxf(fmaxf(tile[ly][8 + 4 * c4], tile[ly + 1][8 + 4 * c4]), tile[ly + 2][8 + 4 * c4]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int x = xb + q;
          // FD:673-676: local_max is 0 on the border ring, the valid 3x3 max inside
          if (x == 0 || y == 0 || x == W - 1 || y == H - 1) {
            r[q] = (hv[q] == 0.f) ? hv[q] : 0.f;
          } else {
            const float m = fmaxf(fmaxf(colmax[q], colmax[q + 1]), colmax[q + 2]);
            r[q] = (hv[q] == m) ? hv[q] : 0.f;
          }
        }
      }
      *reinterpret_cast<float4*>(heat + plane + (long long)y * W + xb) = make_float4(r[0], r[1], r[2], r[3]);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (r[q] > 0.f) atomicAdd(&lhist[score_bin(r[q])], 1u);
    }
  }
  __syncthreads();
  uint32_t* gh = hist + (long long)b * FF3D_HIST_BINS;
  for (int i = tid; i < FF3D_HIST_BINS; i += 256) {
    const uint32_t c = lhist[i];
    if (c) atomicAdd(&gh[i], c);
  }
}

// ------------------------------------------------------------------------------------------------
// Top-k.  One 1024-thread block per sample.
constexpr int TK_THREADS = 1024;
constexpr int TK_CAP = 4096;  // candidates sortable in LDS

__device__ __forceinline__ unsigned long long make_key(float v, unsigned idx) {
  return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xffffffffu - idx);
}

// descending bitonic sort of n2 (power of two) 64-bit keys in LDS
__device__ void bitonic_desc(unsigned long long* keys, int n2) {
  for (int size = 2; size <= n2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < (n2 >> 1); i += blockDim.x) {
        const int pos = 2 * i - (i & (stride - 1));
        const int j = pos + stride;
        const bool up = (pos & size) == 0;
        const unsigned long long a = keys[pos], c = keys[j];
        if ((a < c) == up) {
          keys[pos] = c;
          keys[j] = a;
        }
      }
    }
  }
  __syncthreads();
}

// Threshold bin of a frame: largest t with sum_{i >= t} hist[i] >= k (s_t), the candidate count that implies (s_M), or
// zero mode (fewer than k positive scores: the remaining slots are the lowest-index zeros).  T threads, T | 4096.
template <int T>
__device__ __forceinline__ void topk_threshold(const uint32_t* __restrict__ gh, int k, uint32_t* scan, int* s_t, int* s_zero,
                                               int* s_M) {
  constexpr int BPT = FF3D_HIST_BINS / T;
  const int tid = threadIdx.x;
  uint32_t c[BPT];
  uint32_t mine = 0;
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    c[i] = gh[tid * BPT + i];
    mine += c[i];
  }
  scan[tid] = mine;
  if (tid == 0) *s_t = 0, *s_zero = 0, *s_M = 0;
  __syncthreads();
  for (int off = 1; off < T; off <<= 1) {          // inclusive suffix scan (Hillis-Steele)
    const uint32_t add = (tid + off < T) ? scan[tid + off] : 0u;
    __syncthreads();
    scan[tid] += add;
    __syncthreads();
  }
  const uint32_t incl = scan[tid], above = incl - mine, total_pos = scan[0];
  if (total_pos < (uint32_t)k) {
    if (tid == 0) *s_zero = 1, *s_t = 0, *s_M = (int)total_pos;
  } else if (above < (uint32_t)k && incl >= (uint32_t)k) {
    uint32_t cum = above;
    for (int i = BPT - 1; i >= 0; --i) {
      cum += c[i];
      if (cum >= (uint32_t)k) {
        *s_t = tid * BPT + i;
        *s_M = (int)cum;
        break;
      }
    }
  }
  __syncthreads();
}

// Pass 1, many blocks per frame: block (chunk, b) scans its slice of the frame's score row and appends the candidates
// (score in a bin >= the threshold bin; plus the leading zeros in zero mode) to the frame's candidate list in the
// workspace.  Candidates are staged in LDS and a block reserves its range with ONE atomic on the frame counter.
// (The first version scanned the 1.3 MB row with one block per frame: 87 us at any batch size, 3 x per forward.)
constexpr int TC_THREADS = 256, TC_STAGE = 1024;
__global__ __launch_bounds__(TC_THREADS) void topk_compact_kernel(const float* __restrict__ heat,
                                                                  const uint32_t* __restrict__ hist,
                                                                  unsigned long long* __restrict__ workspace,
                                                                  uint32_t* __restrict__ counters, int n, int k) {
  __shared__ unsigned long long stage[TC_STAGE];
  __shared__ uint32_t scan[TC_THREADS];
  __shared__ int s_t, s_zero, s_M, s_n, s_base;
  const int b = blockIdx.y, tid = threadIdx.x;
  const float* h = heat + (long long)b * n;
  unsigned long long* ws = workspace + (long long)b * n;
  topk_threshold<TC_THREADS>(hist + (long long)b * FF3D_HIST_BINS, k, scan, &s_t, &s_zero, &s_M);
  const int t = s_t, zero_mode = s_zero;
  if (tid == 0) s_n = 0;
  __syncthreads();
  auto push = [&](float v, int i) {
    const bool take = (v > 0.f && score_bin(v) >= t) || (zero_mode && v == 0.f && i < k);
    if (take) {
      const int slot = atomicAdd(&s_n, 1);
      if (slot < TC_STAGE)
        stage[slot] = make_key(v, (unsigned)i);
      else                                             // huge tie group: straight to the global list
        ws[atomicAdd(&counters[b], 1u)] = make_key(v, (unsigned)i);
    }
  };
  if ((n & 3) == 0) {
    const float4* h4 = reinterpret_cast<const float4*>(h);
    const int n4 = n >> 2, per = (n4 + (int)gridDim.x - 1) / (int)gridDim.x;
    const int lo = (int)blockIdx.x * per, hi = min(n4, lo + per);
    constexpr int UN = 4;
    for (int i0 = lo + tid; i0 < hi; i0 += TC_THREADS * UN) {
      float4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + u * TC_THREADS;
        v[u] = i < hi ? h4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + u * TC_THREADS;
        if (i < hi && (v[u].x > 0.f || v[u].y > 0.f || v[u].z > 0.f || v[u].w > 0.f || (zero_mode && 4 * i < k))) {
          push(v[u].x, 4 * i);
          push(v[u].y, 4 * i + 1);
          push(v[u].z, 4 * i + 2);
          push(v[u].w, 4 * i + 3);
        }
      }
    }
  } else {
    const int per = (n + (int)gridDim.x - 1) / (int)gridDim.x, lo = (int)blockIdx.x * per, hi = min(n, lo + per);
    for (int i = lo + tid; i < hi; i += TC_THREADS) push(h[i], i);
  }
  __syncthreads();
  const int m = min(s_n, TC_STAGE);
  if (tid == 0 && m > 0) s_base = (int)atomicAdd(&counters[b], (uint32_t)m);
  __syncthreads();
  for (int i = tid; i < m; i += TC_THREADS) ws[s_base + i] = stage[i];
}

// Pass 2, one 1024-thread block per frame: sort the frame's candidate list (order of arrival is irrelevant: keys are
// distinct) and emit the k best.
__global__ __launch_bounds__(TK_THREADS) void topk_kernel(const uint32_t* __restrict__ counters,
                                                          long long* __restrict__ idx_out,
                                                          unsigned long long* __restrict__ workspace, int n, int k) {
  __shared__ unsigned long long keys[TK_CAP];
  __shared__ uint32_t bytehist[256];
  __shared__ int s_cnt;
  __shared__ unsigned long long s_prefix;
  __shared__ uint32_t s_remaining;

  const int b = blockIdx.x, tid = threadIdx.x;
  unsigned long long* ws = workspace + (long long)b * n;
  int M = (int)counters[b];
  const bool in_lds = M <= TK_CAP;
  if (in_lds)
    for (int i = tid; i < M; i += TK_THREADS) keys[i] = ws[i];
  __syncthreads();

  // ---- 3. rare path: too many candidates for LDS (huge tie groups) - exact radix select of the
  //         k-th largest 64-bit key over the global candidate list, then keep keys >= it.
  if (!in_lds) {
    if (tid == 0) {
      s_prefix = 0ull;
      s_remaining = (uint32_t)k;
    }
    __syncthreads();
    for (int shift = 56; shift >= 0; shift -= 8) {
      if (tid < 256) bytehist[tid] = 0;
      __syncthreads();
      const unsigned long long prefix = s_prefix;
      const unsigned long long himask = (shift == 56) ? 0ull : (~0ull << (shift + 8));
      for (int i = tid; i < M; i += TK_THREADS) {
        const unsigned long long key = ws[i];
        if ((key & himask) == prefix) atomicAdd(&bytehist[(key >> shift) & 0xff], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t rem = s_remaining, cum = 0;
        int byte = 255;
        for (; byte > 0; --byte) {
          if (cum + bytehist[byte] >= rem) break;
          cum += bytehist[byte];
        }
        s_remaining = rem - cum;
        s_prefix = prefix | ((unsigned long long)byte << shift);
      }
      __syncthreads();
    }
    const unsigned long long kth = s_prefix;  // keys are distinct: exactly k keys are >= kth
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int i = tid; i < M; i += TK_THREADS) {
      const unsigned long long key = ws[i];
      if (key >= kth) {
        const int slot = atomicAdd(&s_cnt, 1);
        if (slot < TK_CAP) keys[slot] = key;
      }
    }
    __syncthreads();
    M = min(s_cnt, TK_CAP);
  }

  // ---- 4. sort candidates (score desc, index asc) and emit the first k
  int n2 = 1;
  while (n2 < M) n2 <<= 1;
  if (n2 < 2) n2 = 2;
  for (int i = M + tid; i < n2; i += TK_THREADS) keys[i] = 0ull;
  bitonic_desc(keys, n2);
  for (int j = tid; j < k; j += TK_THREADS) {
    const unsigned lo = (unsigned)(keys[j] & 0xffffffffull);
    idx_out[(long long)b * k + j] = (long long)(0xffffffffu - lo);
  }
}

// ------------------------------------------------------------------------------------------------
// Query gathers + positive-mask update.  One 128-thread block per selected proposal.
__global__ __launch_bounds__(128) void query_gather_kernel(
    const float* __restrict__ feat, const float* __restrict__ heat, const long long* __restrict__ idx,
    const float* __restrict__ cls_w, const float* __restrict__ cls_b, float* __restrict__ qfeat, long long qf_sb,
    long long qf_sq, long long qf_sc, float* __restrict__ qpos, float* __restrict__ qscore,
    long long* __restrict__ qlabel, float* __restrict__ mask, int C, int K, int H, int W, int k, int q_offset,
    int Nq, int mask_mode, int nms_kernel, uint32_t small_bits) {
  const int b = blockIdx.x / k, j = blockIdx.x - b * k;
  const int HW = H * W;
  const long long flat = idx[(long long)b * k + j];
  const int cls = (int)(flat / HW), cell = (int)(flat - (long long)cls * HW);
  const int y = cell / W, x = cell - y * W;
  const int q = q_offset + j;
  const int tid = threadIdx.x;

  if (feat && qfeat) {
    const float* f = feat + (long long)b * C * HW + cell;
    float* o = qfeat + b * qf_sb + q * qf_sq;
    for (int c = tid; c < C; c += 128) o[c * qf_sc] = f[(long long)c * HW] + (cls_w[c * K + cls] + cls_b[c]);
  }
  if (qscore) {
    for (int c = tid; c < K; c += 128) qscore[((long long)b * K + c) * Nq + q] = heat[((long long)b * K + c) * HW + cell];
  }
  if (tid == 0) {
    if (qpos) {
      qpos[((long long)b * Nq + q) * 2 + 0] = (float)x + 0.5f;
      qpos[((long long)b * Nq + q) * 2 + 1] = (float)y + 0.5f;
    }
    if (qlabel) qlabel[(long long)b * Nq + q] = cls;
  }
  if (mask && mask_mode) {
    // FD:725-782: scatter 1 at the proposal (class plane `cls`, or every class in 'pos' mode), 3x3
    // max-pool dilation (pad 1) except for the kernel-1 classes, acc *= (1 - dilated)  ==  clear.
    const int c_lo = (mask_mode == 2) ? 0 : cls, c_hi = (mask_mode == 2) ? K : cls + 1;
    const int items = (c_hi - c_lo) * 9;
    for (int i = tid; i < items; i += 128) {
      const int c = c_lo + i / 9, d = i % 9;
      const int dy = d / 3 - 1, dx = d % 3 - 1;
      const bool small = ((small_bits >> c) & 1u) || nms_kernel == 1;
      if (small && (dy || dx)) continue;
      const int yy = y + dy, xx = x + dx;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      mask[((long long)b * K + c) * HW + yy * W + xx] = 0.f;
    }
  }
}

// Zero fill of the histogram / counters as a KERNEL, not hipMemsetAsync (round 6).  A hipMemsetAsync captured into a hipGraph becomes
// a memset node; on ROCm 7.2 (graph packet capture on, the default) the captured head then faulted the GPU on the replay that follows
// [replay, any eager launch, hipDeviceSynchronize]: "Memory access fault" on an address outside every allocation of the process's
// allocator - memory of the runtime's own.  tools/bisect_graph_fault.py: of ten launch families only the one with memset nodes
// faults; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 makes the same graph safe; smallest faulting capture = one heatmap_nms + one topk call
// (profiles/r06_a_graph_fault_bisect.txt, r06_b_graph_memset_repro.txt).  FF3D_MEMSET_NODES=1 restores the memset nodes (the A/B hook
// of those records).
__global__ __launch_bounds__(256) void zero_u32_kernel(uint32_t* __restrict__ p, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0u;
}

int zero_u32(uint32_t* p, long long n, hipStream_t s) {
  static const bool memset_nodes = [] {
    const char* e = getenv("FF3D_MEMSET_NODES");
    return e && e[0] == '1';
  }();
  if (memset_nodes) return hipMemsetAsync(p, 0, (size_t)n * sizeof(uint32_t), s) == hipSuccess ? FF3D_OK : FF3D_ERR_LAUNCH;
  hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n);
  return FF3D_OK;
}

}  // namespace

extern "C" int ff3d_heatmap_nms(const float* logits, const float* logits_b, const float* mask_in, float* mask_next,
                                float* heat, uint32_t* hist, int B, int K, int H, int W, int nms_kernel,
                                uint32_t small_class_bits, ff3d_stream_t stream) {
  FF3D_REQUIRE(logits && heat && hist, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && K > 0 && K <= 32 && H > 0 && W > 0 && B <= 65535 && K <= 65535, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(nms_kernel == 1 || nms_kernel == 3, FF3D_ERR_UNSUPPORTED);
  FF3D_REQUIRE(nms_kernel == 1 || (H >= 3 && W >= 3), FF3D_ERR_BAD_SHAPE);
  hipStream_t s = static_cast<hipStream_t>(stream);
  ff3d_clear_error();
  if (zero_u32(hist, (long long)B * FF3D_HIST_BINS, s) != FF3D_OK) return FF3D_ERR_LAUNCH;
  static const bool no_wide = [] {                                       // A/B hook: FF3D_NMS_WIDE=0 = the 32 x 8-tile kernel everywhere
    const char* e = getenv("FF3D_NMS_WIDE");
    return e && e[0] == '0';
  }();
  const bool wide = !no_wide && (W & 3) == 0 && ((long long)H * W & 3) == 0 && ff3d_aligned16(logits) && ff3d_aligned16(heat) &&
                    (!logits_b || ff3d_aligned16(logits_b)) && (!mask_in || ff3d_aligned16(mask_in)) &&
                    (!mask_next || ff3d_aligned16(mask_next));
  if (wide) {
    hipLaunchKernelGGL(heatmap_nms_wide_kernel, dim3((H + NV_SY - 1) / NV_SY, K, B), dim3(256), 0, s, logits, logits_b, mask_in,
                       mask_next, heat, hist, K, H, W, nms_kernel, small_class_bits);
    return ff3d_launch_status();
  }
  const int strips = (H + TY - 1) / TY;
  hipLaunchKernelGGL(heatmap_nms_kernel, dim3(strips, K, B), dim3(256), 0, s, logits, logits_b, mask_in, mask_next,
                     heat, hist, K, H, W, nms_kernel, small_class_bits);
  return ff3d_launch_status();
}

extern "C" size_t ff3d_topk_workspace_bytes(int B, int n) {
  if (B <= 0 || n <= 0) return 0;
  // per frame: n 64-bit candidate keys (worst case: every score ties) + one counter (8 bytes each, keeps the alignment)
  return ((size_t)B * (size_t)n + (size_t)B) * sizeof(unsigned long long);
}

extern "C" int ff3d_topk(const float* heat, const uint32_t* hist, int64_t* idx_out, void* workspace, int B, int n,
                         int k, ff3d_stream_t stream) {
  FF3D_REQUIRE(heat && hist && idx_out && workspace, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && n > 0 && k >= 1 && k <= TK_CAP && k <= n, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE((n & 3) != 0 || ff3d_aligned16(heat), FF3D_ERR_ALIGNMENT);
  ff3d_clear_error();
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned long long* ws = reinterpret_cast<unsigned long long*>(workspace);
  uint32_t* counters = reinterpret_cast<uint32_t*>(ws + (size_t)B * (size_t)n);
  if (zero_u32(counters, 2ll * B, s) != FF3D_OK) return FF3D_ERR_LAUNCH;
  // enough chunks to put ~2 blocks on every CU whatever the batch size (the scan is a pure streaming read)
  int chunks = (512 + B - 1) / B;
  if (chunks > 64) chunks = 64;
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(topk_compact_kernel, dim3(chunks, B), dim3(TC_THREADS), 0, s, heat, hist, ws, counters, n, k);
  hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(TK_THREADS), 0, s, counters, reinterpret_cast<long long*>(idx_out), ws, n, k);
  return ff3d_launch_status();
}

extern "C" int ff3d_query_gather(const float* feat, const float* heat, const int64_t* idx, const float* cls_w,
                                 const float* cls_b, float* qfeat, int64_t qf_sb, int64_t qf_sq, int64_t qf_sc,
                                 float* qpos, float* qscore, int64_t* qlabel, float* mask, int B, int C, int K, int H,
                                 int W, int k, int q_offset, int Nq, int mask_mode, int nms_kernel,
                                 uint32_t small_class_bits, ff3d_stream_t stream) {
  FF3D_REQUIRE(idx, FF3D_ERR_NULL);
  FF3D_REQUIRE(!(feat && qfeat) || (cls_w && cls_b), FF3D_ERR_NULL);
  FF3D_REQUIRE(!qscore || heat, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && C > 0 && K > 0 && K <= 32 && H > 0 && W > 0 && k > 0 && q_offset >= 0 && q_offset + k <= Nq,
               FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(mask_mode >= 0 && mask_mode <= 2, FF3D_ERR_UNSUPPORTED);
  FF3D_REQUIRE(nms_kernel == 1 || nms_kernel == 3, FF3D_ERR_UNSUPPORTED);
  ff3d_clear_error();
  hipLaunchKernelGGL(query_gather_kernel, dim3(B * k), dim3(128), 0, static_cast<hipStream_t>(stream), feat, heat,
                     reinterpret_cast<const long long*>(idx), cls_w, cls_b, qfeat, (long long)qf_sb, (long long)qf_sq,
                     (long long)qf_sc, qpos, qscore, reinterpret_cast<long long*>(qlabel), mask, C, K, H, W, k,
                     q_offset, Nq, mask_mode, nms_kernel, small_class_bits);
  return ff3d_launch_status();
}
