// fp32-class dense work on the fp16 matrix cores of gfx950: 3x3 convolution (implicit GEMM) and plain GEMM with every
// fp32 operand carried as a (hi, lo) pair of fp16 values and three MFMA passes per product,
//     x*w ~= x_hi*w_hi + (x_hi*w_lo' + x_lo'*w_hi) * 2^-11,      x_hi = fp16(x),  x_lo' = fp16((x - x_hi) * 2^11)
// accumulated in fp32 (two accumulators: the main term and the scaled cross terms).  fp16 x fp16 products are exact in
// the fp32 accumulator and the dropped lo*lo term is below 2^-22 relative, so the result is within a few fp32 ulps of an
// fp32 FMA chain - while v_mfma_f32_16x16x32_f16 runs at 16x the rate of the fp32-input MFMA (2.5 PFLOP/s vs 157
// TFLOP/s dense), i.e. ~5x the fp32 peak after paying for the three passes.  The scaling of the low parts keeps them in
// fp16's normal range (an unscaled x_lo of a weight ~0.03 would be subnormal).
//
// This is the MI355X answer to the dense layers of the FocalDecoder head (heatmap / pyramid 3x3 convs FD:150-162,
// 202-229; value_proj / roi_mlp GEMMs), which in MIOpen / hipBLASLt fp32 sit at the 157 TFLOP/s fp32-MFMA ceiling.
//
// Kernel: 128x128 output tile per 256-thread block (4 waves as 2x2, 64x64 per wave = 4x4 MFMA tiles x 2 accumulators),
// K-step 32 (one MFMA K), operand tiles A_hi/A_lo/B_hi/B_lo streamed global -> LDS with 16-byte global_load_lds into a
// double buffer (64 KiB), one barrier per K-step, the next step's loads in flight under the current step's 48 MFMAs per
// wave.  LDS rows are 64 bytes (4 chunks of 16 B) with the chunk index XOR-swizzled by f((row >> 2) & 3), f = {0, 2, 3, 1}
// - applied on the per-lane SOURCE address (the DMA destination is lane-linear) and again on the fragment read.  f is
// chosen for the hardware's ds_read_b128 service groups (lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...): each
// group then touches 16 distinct 16-byte bank columns (a plain row-index XOR leaves every group 2-way conflicted).
// Tried and rejected (measured on the 256-channel conv, 3.50 ms): spreading the next step's DMA issues between groups of
// 6 MFMAs with sched_barrier pins (4.27 ms); the 256x128 / 8-wave / triple-buffered variant below (3.55 ms; kept as a
// template instance); v_mfma_f32_32x32x16_f16 tiles (3.77 ms); 64x128 two-wave blocks (6.7 ms); 128x128 triple-buffered
// (4.08 ms); for the short-K value_proj GEMM an A-stationary kernel (A strip resident in LDS, weights streamed; 2.38 vs
// 2.54 ms alone, no gain inside the head); A fragments loaded straight from global memory into registers, only the weights
// through the LDS DMA (3.7 vs 2.6 ms: fragment-shaped 16-byte row loads cost more than the DMA pieces they replace);
// persistent blocks walking the tiles so a tile's stores could drain under the next tile's work (2.8 vs 2.6 ms: the
// next tile's first vmcnt(0) waits for the stores anyway - loads and stores share the in-order VM counter).  What did pay is cutting the DMA pieces per MFMA: convhalo.hip.  All sit at ~1.05 PFLOP/s of MFMA work = 56 % MFMA-busy at
// the ~1.78 GHz the chip sustains under this load (PMC: no LDS bank conflicts, LDS 19 % busy, VALU:MFMA 0.5).  Implicit GEMM: the A row of output pixel m for K-step ks is the
// 64-byte channel run [c0, c0+32) of input pixel (y*stride + dy - 1, x*stride + dx - 1) of an NHWC fp16 tensor, or the
// zero row every operand plane carries after its last real row (padding / ragged M, N).  DMA addresses are an SGPR plane
// base + a 32-bit per-lane byte offset; per K-step a lane adds wave-uniform displacements only (first version: 64-bit
// pointer arithmetic + a tap division per issue = 2.5 VALU instructions per MFMA, PMC).  Blocks are XCD-remapped so neighbouring pixel tiles share an L2.
#include <cstdlib>

#include <type_traits>

#include "ff3d_common.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int SM_BN = 128, SM_BK = 32;
constexpr float SM_LO_SCALE = 2048.f, SM_LO_INV = 1.f / 2048.f;

__device__ __forceinline__ int sm_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

struct SplitMMParams {
  const _Float16 *a_hi, *a_lo, *w_hi, *w_lo;     // every plane ends with one zero row (a pixel / a K-row): the padding source
  const float* bias;
  float* out;
  _Float16 *out_hi, *out_lo;    // out_mode 2: the result as a (hi, lo') pair, rows of N (NHWC for a conv)
  const _Float16 *res_hi, *res_lo;   // optional residual pair (M, N) added before the activation
  float upper;                  // activation clamp: min(relu(v), upper) (6 for ReLU6; +inf otherwise)
  int M, N, K;                  // conv: M = B*Ho*Wo, K = 9*C
  int conv, C, H, W, Ho, Wo, stride;
  int relu, out_mode;           // 0: (M, N) fp32 row-major, 1: NCHW fp32 (conv), 2: (M, N) split fp16 pair,
                                // 3 (one-plane bf16 instances only): (M, N) bf16 row-major at `out_hi`
  int ksplit;                   // GEMM only: gridDim.y K-slices, slice s writes its raw partial sums to plane s of `out`
                                // (= the (ksplit, M, N) workspace); splitk_reduce_kernel adds the planes in order
  unsigned a_zero, b_zero;      // byte offsets of the zero rows
  Ff3dScale sc;                 // range normalisation (ff3d.h): operand exponents in, output exponent out
  // periodic GEMM (ff3d_gemm_f16x3_rowbias): M = nbatch frames of `period` rows; tiles never straddle frames and are walked
  // frame-fastest, so the (period, N) bias table tile of a row block is reused by all frames while it is still in L2
  int period, nbatch;
  const float* bias_tab;
  // round 5, split-K GEMM with few output columns (roi_mlp.0: N = 512): the launcher passes the operands SWAPPED (A := the weight,
  // W := the activation), so that one block covers 256 of the N weight rows and the activation panel is streamed by N / 256 blocks
  // instead of N / 128; the raw partial sums are then stored transposed - plane[col * M + row] - which is the (rows, N) layout the
  // reduce kernel expects.  Sibling blocks (the same activation tile) are adjacent in the grid.
  int swap_out;
  // round 6, stride-2 3x3 conv with few output channels (the BEV pyramid, FD:150-162: N = 256): operands SWAPPED for the conv too
  // (CONVB instance): A := the weight (one block covers all 256 output channels), B := the activation with the implicit-GEMM gather
  // (128 output pixels per block), so that every activation tile is gathered ONCE instead of once per 128-channel tile; here M = the
  // output channels, N = B * Ho * Wo pixels, a_zero / b_zero follow the operands, out_mode 1 (NCHW fp32), transposed accumulators.
  int conv_b;
};

// 16-byte LDS-DMA with the address as SGPR base + 32-bit per-lane byte offset (no 64-bit VALU arithmetic per issue)
__device__ __forceinline__ void glds16(const _Float16* base, unsigned byte_off, _Float16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(base) + byte_off,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// WM = waves along M (block = WM x 2 waves, tile = 64*WM x 128), NBUF = LDS pipeline depth.
//   <2, 2>: 128x128 tile, 256 threads, 64 KiB, two blocks per CU, DMA one K-step ahead, __syncthreads per step.
//   <4, 3>: 256x128 tile, 512 threads, 144 KiB, one block per CU, DMA two K-steps ahead: counted s_waitcnt vmcnt(6)
//           (the newest step stays in flight across the barrier) + raw s_barrier.
//   <2, 4>: round 3, the SMALL-GRID form: 128x128 tile, 128 KiB, one block per CU, DMA THREE K-steps ahead - used when the
//           grid is at most 256 blocks (one per CU anyway: the pyramid convs at 1 - 4 frames), where nothing else on the CU
//           covers a block's global -> LDS round trips: 75 vs 89 us (90x90 -> 45x45, 4 frames), 81 vs 94 (180x180 -> 90x90, 1
//           frame).  From 257 to 512 blocks two co-resident <2, 2> blocks per CU are faster than two rounds of this form
//           (123 vs 163 us, 331 vs 391 us for roi_mlp.0 at 4 frames: profiles/r03_q_deep_ab.txt).
// TR: accumulate the TRANSPOSED tile (the MFMA's A / B fragment layouts are symmetric, so swapping the two operands yields
// D^T): a lane then holds 4 consecutive output COLUMNS n of one row m instead of 4 consecutive rows of one column - what
// the row-major outputs want (GEMM fp32 (M, N): one 16-byte store instead of four 4-byte stores 64 B apart; NHWC pair
// planes: 8-byte stores, no lane exchange).  PMC WRITE_SIZE of the value_proj launch was 4.87 GB for 4.18 GB of output
// with the scalar stores.  The NCHW conv output (4 consecutive pixels per lane) keeps TR = false.
// PL (round 5): operand planes.  2 = the split-fp16 arithmetic above.  1 = ONE bf16 plane per operand on v_mfma_f32_16x16x32_bf16
// (BASELINE configs[4], "bf16 QKV/FFN on MFMA": roi_mlp.0 reading the bf16 RoI matrix): exact products, fp32 accumulation, bias in
// fp32, ONE rounding of the result to bf16, ReLU on the rounded value (oracle/ff3d_oracle.py lin(lowp=True)); result as fp32
// (out_mode 0) or bf16 rows (out_mode 3); a_lo / w_lo unused, no exponents.
template <int WM, int NBUF, bool TR, int PL = 2, bool CONVB = false>
__global__ __launch_bounds__(WM * 128, (NBUF == 2 && WM <= 2) ? (WM == 1 ? 3 : 2) : 1) void splitmm_kernel(SplitMMParams p) {
  static_assert(!CONVB || (TR && PL == 2), "the swapped conv stores NCHW fp32 from transposed accumulators");
  constexpr int T = WM * 128, BM = WM * 64;
  constexpr int A_TILE = BM * SM_BK, B_TILE = SM_BN * SM_BK;      // halves per operand plane tile
  constexpr int BUF = PL * (A_TILE + B_TILE);                      // halves per pipeline stage
  constexpr int BJ = (SM_BN * 4) / T;                              // B slots per thread (2 or 1)
  constexpr int PIECES = PL * (2 + BJ);                            // DMA instructions per thread per K-step
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];   // [NBUF][A_hi | A_lo | B_hi | B_lo]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_tiles = (p.N + SM_BN - 1) / SM_BN;
  const int m_tiles = p.period ? p.nbatch * ((p.period + BM - 1) / BM) : (p.M + BM - 1) / BM;
  const unsigned lid = ff3d_xcd_remap(blockIdx.x, (unsigned)(n_tiles * m_tiles));
  int n0 = (int)(lid % n_tiles) * SM_BN;
  int m0 = (int)(lid / n_tiles) * BM, m_end = p.M, row0 = 0;     // rows [m0, m_end) are real; row0 = m0's row in its frame
  if (p.swap_out) n0 = (int)(lid / m_tiles) * SM_BN, m0 = (int)(lid % m_tiles) * BM;
  if (p.period) {
    const int mt = (int)(lid / n_tiles), t = mt / p.nbatch, b = mt - t * p.nbatch;
    row0 = t * BM, m0 = b * p.period + row0, m_end = (b + 1) * p.period;
  }

  // ---- staging geometry: thread owns slots s = j*T + tid of every tile: row s>>2, swizzled chunk s&3.
  // Per slot: byte offset of the row's data for the centre tap (+ chunk), of the zero row (+ chunk), and the taps that
  // read real data; per K-step only wave-uniform (scalar) displacements are added.
  unsigned a_c[2], a_z[2], a_valid[2], b_c[BJ], b_z[CONVB ? BJ : 1], b_valid[CONVB ? BJ : 1];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int s = j * T + tid, row = s >> 2;
    const unsigned chunk_b = (unsigned)(((s & 3) ^ sm_swz(row)) * 16);   // source chunk whose data lands in LDS slot s
    const int m = m0 + row;
    a_z[j] = p.a_zero + chunk_b;
    a_valid[j] = 0;
    a_c[j] = a_z[j];
    if (m < m_end) {
      if (!p.conv) {
        a_c[j] = (unsigned)m * (unsigned)p.K * 2u + chunk_b;
        a_valid[j] = 1;
      } else {
        const int hw = p.Ho * p.Wo, b = m / hw, r = m - b * hw, yo = r / p.Wo, xo = r - yo * p.Wo;
        const int yc = yo * p.stride, xc = xo * p.stride;                // centre tap: always inside the image
        a_c[j] = (unsigned)(((b * p.H + yc) * p.W + xc) * p.C) * 2u + chunk_b;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int y = yc + t / 3 - 1, x = xc + t % 3 - 1;
          if (y >= 0 && y < p.H && x >= 0 && x < p.W) a_valid[j] |= 1u << t;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int s = j * T + tid, row = s >> 2, n = n0 + row;
    const unsigned chunk_b = (unsigned)(((s & 3) ^ sm_swz(row)) * 16);
    if (CONVB) {                                     // B row n = output pixel n: centre-tap address + the taps that read real data
      b_z[j] = p.b_zero + chunk_b;
      b_valid[j] = 0;
      b_c[j] = b_z[j];
      if (n < p.N) {
        const int hw_o = p.Ho * p.Wo, b = n / hw_o, r = n - b * hw_o, yo = r / p.Wo, xo = r - yo * p.Wo;
        const int yc = yo * p.stride, xc = xo * p.stride;
        b_c[j] = (unsigned)(((b * p.H + yc) * p.W + xc) * p.C) * 2u + chunk_b;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int y = yc + t / 3 - 1, x = xc + t % 3 - 1;
          if (y >= 0 && y < p.H && x >= 0 && x < p.W) b_valid[j] |= 1u << t;
        }
      }
    } else {
      b_c[j] = (n < p.N ? (unsigned)n * (unsigned)p.K * 2u : p.b_zero) + chunk_b;
    }
  }
  // wave-uniform K-step state, advanced incrementally (no division in the loop); stage() is called in K order
  int st_tap = 0, st_dy = 0, st_dx = 0, st_c0 = 0;
  auto stage = [&](int ks, int buf) {
    // displacement of this K-step relative to the per-slot base: conv = tap shift + channel run, GEMM = ks * 64 bytes
    const int s_k = p.conv ? st_c0 * 2 : ks * (SM_BK * 2);
    const int s_tap = (p.conv || CONVB) ? ((st_dy - 1) * p.W + (st_dx - 1)) * p.C * 2 : 0;
    const int tap = st_tap;
    _Float16* base = lds + buf * BUF;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      _Float16* dst = base + (j * T + wave * 64) * 8;                    // wave-uniform; the DMA adds lane*16 B
      // (CONVB: the A operand is the weight, plain rows - the tap state belongs to the B operand)
      const unsigned ao = CONVB ? a_c[j] + (unsigned)s_k
                                : (((a_valid[j] >> tap) & 1u) ? a_c[j] + (unsigned)s_tap : a_z[j]) + (unsigned)s_k;
      glds16(p.a_hi, ao, dst);
      if (PL == 2) glds16(p.a_lo, ao, dst + A_TILE);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      _Float16* dst = base + PL * A_TILE + (j * T + wave * 64) * 8;
      const unsigned bo = CONVB ? (((b_valid[j] >> tap) & 1u) ? b_c[j] + (unsigned)s_tap : b_z[j]) + (unsigned)(st_c0 * 2)
                                : b_c[j] + (unsigned)(ks * (SM_BK * 2));
      glds16(p.w_hi, bo, dst);
      if (PL == 2) glds16(p.w_lo, bo, dst + B_TILE);
    }
    if (p.conv || CONVB) {
      st_c0 += SM_BK;
      if (st_c0 == p.C) {
        st_c0 = 0, ++st_tap, ++st_dx;
        if (st_dx == 3) st_dx = 0, ++st_dy;
      }
    }
  };

  const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, kq = lane >> 4;
  f32x4 acc_m[4][4], acc_x[4][4];                 // main term, scaled cross terms
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc_m[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}, acc_x[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // periodic GEMM: the (row, n) bias table enters as the INITIAL accumulator value (divided by the power-of-two operand
  // scale, exact), so its 64 scattered loads per lane overlap the pipeline fill instead of sitting in the epilogue
  // (there they cost 2.7 ms of an 8.3 ms launch, measured)
  if (p.bias_tab) {
    const float inv = ff3d_pow2(-(ff3d_ld_exp(p.sc.a_exp) + ff3d_ld_exp(p.sc.w_exp)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + fr;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // element r of the accumulator: (row kq*4 + r, column fr) of the 16x16 tile, transposed when TR
          const int row = wr * 64 + i * 16 + (TR ? fr : kq * 4 + r);          // row inside the block tile
          const int nn = TR ? n0 + wc * 64 + j * 16 + kq * 4 + r : n;
          if (nn < p.N && m0 + row < m_end) acc_m[i][j][r] = p.bias_tab[(long long)(row0 + row) * p.N + nn] * inv;
        }
    }
  }

  // fragment read offsets (halves) inside an operand tile: row*32 + swizzled chunk*8
  int a_rd[4], b_rd[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wr * 64 + i * 16 + fr, rb = wc * 64 + i * 16 + fr;
    a_rd[i] = ra * SM_BK + ((kq ^ sm_swz(ra)) * 8);
    b_rd[i] = rb * SM_BK + ((kq ^ sm_swz(rb)) * 8);
  }

  // K range of this block (split-K GEMM: slice blockIdx.y of gridDim.y; stage() takes absolute step numbers)
  const int nk_all = p.K / SM_BK, per = (nk_all + (int)gridDim.y - 1) / (int)gridDim.y;
  const int k_lo = (int)blockIdx.y * per, nk = min(nk_all, k_lo + per);
  if ((p.conv || CONVB) && k_lo > 0) {            // split-K conv (round 6: ff3d_conv3x3_f16x3_splitk): the slice starts inside the tap walk
    st_tap = (k_lo * SM_BK) / p.C, st_c0 = k_lo * SM_BK - st_tap * p.C;
    st_dy = st_tap / 3, st_dx = st_tap - st_dy * 3;
  }
  constexpr int PF = NBUF - 1;                    // K-steps the DMA runs ahead
  if (k_lo < nk) stage(k_lo, 0);
#pragma unroll
  for (int q = 1; q < PF; ++q)
    if (NBUF >= 3 && k_lo + q < nk) stage(k_lo + q, q);
  int cur = 0;                                    // buffer of K-step ks
  for (int ks = k_lo; ks < nk; ++ks) {
    if (NBUF == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                            // tile ks landed for every wave; the other buffer is free again
      if (ks + 1 < nk) stage(ks + 1, cur ^ 1);
    } else {
      // this wave's pieces of tile ks have landed once at most the pieces of the younger tiles (ks+1 .. ks+PF-1, as far as
      // they exist) are outstanding (in-order counter)
      const int younger = min(PF - 1, nk - 1 - ks);
      if (younger >= 2)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
      else if (younger == 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();               // ... and every other wave's; all reads of tile ks-1 are retired
      asm volatile("" ::: "memory");
      if (ks + PF < nk) stage(ks + PF, cur == 0 ? NBUF - 1 : cur - 1);   // buffer (ks + PF) % NBUF = the one tile ks-1 used
    }
    const _Float16* t = lds + cur * BUF;
    if (PL == 1) {
      bf16x8 ab[4], bb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ab[i] = *reinterpret_cast<const bf16x8*>(t + a_rd[i]);
        bb[i] = *reinterpret_cast<const bf16x8*>(t + A_TILE + b_rd[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc_m[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[j], ab[i], acc_m[i][j], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[i], bb[j], acc_m[i][j], 0, 0, 0);
      cur = (NBUF == 2) ? (cur ^ 1) : (cur == NBUF - 1 ? 0 : cur + 1);
      continue;
    }
    half8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const half8*>(t + a_rd[i]);
      al[i] = *reinterpret_cast<const half8*>(t + A_TILE + a_rd[i]);
      bh[i] = *reinterpret_cast<const half8*>(t + 2 * A_TILE + b_rd[i]);
      bl[i] = *reinterpret_cast<const half8*>(t + 2 * A_TILE + B_TILE + b_rd[i]);
    }
    // PASS-MAJOR order (round 3, see convhalo.hip): the hi*hi pass over the 16 tiles, then the two cross passes - the two
    // dependent acc_x MFMAs of a tile are 16 instructions apart instead of back to back (a dependent v_mfma_f32_16x16x32_f16
    // waits out the producer's 8 passes: the 56 % MFMA-busy ceiling of rounds 1-2).
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
    cur = (NBUF == 2) ? (cur ^ 1) : (cur == NBUF - 1 ? 0 : cur + 1);
  }

  // ---- epilogue: D row = (lane>>4)*4 + r (output row m), col = lane&15 (output column n)
  const int hw = p.Ho * p.Wo;
  // range normalisation: the accumulators hold sums of SCALED operands; one power of two restores real units, and a pair
  // output gets its own exponent from the guaranteed bound of the layer (ff3d_common.h)
  const int e_a = ff3d_ld_exp(p.sc.a_exp);
  const float sc_in = ff3d_pow2(e_a + ff3d_ld_exp(p.sc.w_exp));
  float sc_out = 1.f, sc_res = 1.f;
  if (p.sc.out_exp) {
    const int e_out = ff3d_out_exp(p.sc, e_a, p.res_hi != nullptr, p.relu ? p.upper : INFINITY);
    if (p.out_mode == 2) sc_out = ff3d_pow2(-e_out);
    if (lid == 0 && blockIdx.y == 0 && tid == 0) *p.sc.out_exp = e_out;
  }
  if (p.res_hi) sc_res = ff3d_pow2(ff3d_ld_exp(p.sc.res_exp));
  if (TR && CONVB) {
    // swapped conv: row m = output channel, columns n .. n + 3 = four consecutive output pixels -> one 16-byte store into the
    // channel's NCHW plane
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wr * 64 + i * 16 + fr;
      if (m >= m_end) continue;
      const float bm = p.bias ? p.bias[m] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + kq * 4;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fmaf(acc_m[i][j][r] + acc_x[i][j][r] * SM_LO_INV, sc_in, bm);
          if (p.relu) v[r] = fminf(fmaxf(v[r], 0.f), p.upper);
        }
        if (n + 3 < p.N && (hw & 3) == 0) {          // 4 consecutive pixels of one image plane (n % 4 == 0)
          const int b = n / hw, q = n - b * hw;
          *reinterpret_cast<float4*>(p.out + ((long long)b * p.M + m) * hw + q) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) {
              const int b = (n + r) / hw, q = (n + r) - b * hw;
              p.out[((long long)b * p.M + m) * hw + q] = v[r];
            }
        }
      }
    }
    return;
  }
  if (TR) {
    // lane: row m = ... + fr, columns n .. n + 3 (n = ... + kq * 4): row-major outputs (out_mode 0 / 2, split-K planes)
    const bool n4 = (p.N & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wr * 64 + i * 16 + fr;
      if (m >= m_end) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + kq * 4;
        if (n >= p.N) continue;
        const long long o = (long long)m * p.N + n;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc_m[i][j][r] + acc_x[i][j][r] * SM_LO_INV;
        if (p.ksplit > 1) {               // raw partial sums of this K slice
          float* plane = p.out + (long long)blockIdx.y * p.M * p.N;
          if (n4) {
            *reinterpret_cast<float4*>(plane + o) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) plane[o + r] = v[r];
          }
          continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float bj = (p.bias && n + r < p.N) ? p.bias[n + r] : 0.f;
          v[r] = fmaf(v[r], sc_in, bj);
          if (p.res_hi && n + r < p.N) v[r] = fmaf((float)p.res_hi[o + r] + (float)p.res_lo[o + r] * SM_LO_INV, sc_res, v[r]);
          if (PL == 1) v[r] = (float)(__bf16)v[r];
          if (p.relu) v[r] = fminf(fmaxf(v[r], 0.f), p.upper);
        }
        if (PL == 1 && p.out_mode == 3) {  // bf16 rows: 4 consecutive columns = one 8-byte store
          __bf16 q[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          __bf16* ob = reinterpret_cast<__bf16*>(p.out_hi);
          if (n4) {
            *reinterpret_cast<uint2*>(ob + o) = *reinterpret_cast<uint2*>(q);
          } else {
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) ob[o + r] = q[r];
          }
        } else if (p.out_mode == 2) {            // (hi, lo') planes, rows of N: 4 consecutive channels = one 8-byte store per plane
          _Float16 h[4], l[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float vs = v[r] * sc_out;
            h[r] = (_Float16)vs;
            l[r] = (_Float16)((vs - (float)h[r]) * SM_LO_SCALE);
          }
          if (n4) {
            *reinterpret_cast<uint2*>(p.out_hi + o) = *reinterpret_cast<uint2*>(h);
            *reinterpret_cast<uint2*>(p.out_lo + o) = *reinterpret_cast<uint2*>(l);
          } else {
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) p.out_hi[o + r] = h[r], p.out_lo[o + r] = l[r];
          }
        } else if (n4) {
          *reinterpret_cast<float4*>(p.out + o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) p.out[o + r] = v[r];
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
      const int mb = m0 + wr * 64 + i * 16 + kq * 4;
      if (p.ksplit > 1) {               // raw partial sums of this K slice (scaling / bias / activation: splitk_reduce_kernel)
        float* plane = p.out + (long long)blockIdx.y * p.M * p.N;
        if (p.swap_out) {               // swapped operands: this lane's 4 consecutive rows are 4 consecutive COLUMNS of the caller's row n
          if (mb + 3 < m_end) {
            *reinterpret_cast<float4*>(plane + (long long)n * p.M + mb) =
                make_float4(acc_m[i][j][0] + acc_x[i][j][0] * SM_LO_INV, acc_m[i][j][1] + acc_x[i][j][1] * SM_LO_INV,
                            acc_m[i][j][2] + acc_x[i][j][2] * SM_LO_INV, acc_m[i][j][3] + acc_x[i][j][3] * SM_LO_INV);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (mb + r < m_end) plane[(long long)n * p.M + mb + r] = acc_m[i][j][r] + acc_x[i][j][r] * SM_LO_INV;
          }
          continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (mb + r < m_end) plane[(long long)(mb + r) * p.N + n] = acc_m[i][j][r] + acc_x[i][j][r] * SM_LO_INV;
        continue;
      }
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = fmaf(acc_m[i][j][r] + acc_x[i][j][r] * SM_LO_INV, sc_in, bj);
        if (p.res_hi && mb + r < m_end) {
          const long long o = (long long)(mb + r) * p.N + n;
          v[r] = fmaf((float)p.res_hi[o] + (float)p.res_lo[o] * SM_LO_INV, sc_res, v[r]);
        }
        if (PL == 1) v[r] = (float)(__bf16)v[r];
        if (p.relu) v[r] = fminf(fmaxf(v[r], 0.f), p.upper);
      }
      if (p.out_mode == 2) {
        // (hi, lo') NHWC planes for a following split-fp16 layer.  Lanes 2k / 2k+1 hold neighbouring columns of the same
        // 4 rows: they swap two rows each so that every lane stores two 4-byte column pairs instead of four halves.
        const bool odd = lane & 1;
        unsigned hs[4], ls[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float vs = v[r] * sc_out;
          const _Float16 h = (_Float16)vs;
          const _Float16 l = (_Float16)((vs - (float)h) * SM_LO_SCALE);
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
          if (mb + r0 + q < m_end) {
            const long long o = (long long)(mb + r0 + q) * p.N + nc;
            *reinterpret_cast<unsigned*>(p.out_hi + o) = ph;
            *reinterpret_cast<unsigned*>(p.out_lo + o) = pl;
          }
        }
      } else if (p.out_mode == 1) {
        if (mb + 3 < m_end && (hw & 3) == 0) {      // 4 consecutive pixels of one image plane
          const int b = mb / hw, q = mb - b * hw;
          *reinterpret_cast<float4*>(p.out + ((long long)b * p.N + n) * hw + q) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (mb + r < m_end) {
              const int b = (mb + r) / hw, q = (mb + r) - b * hw;
              p.out[((long long)b * p.N + n) * hw + q] = v[r];
            }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (mb + r < m_end) p.out[(long long)(mb + r) * p.N + n] = v[r];
      }
    }
  }
}

// Second half of a split-K GEMM: out[m, n] = act(sum_s ws[s, m, n] * 2^(e_a + e_w) + bias[n]), planes added in slice order
// (deterministic - unlike atomics - for any number of slices).  float4 per thread; memory-bound and tiny next to the GEMM.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                            float* __restrict__ out, long long MN, int N, int S, int relu,
                                                            float upper, Ff3dScale sc, int round_bf16 = 0) {
  const float sc_in = ff3d_pow2(ff3d_ld_exp(sc.a_exp) + ff3d_ld_exp(sc.w_exp));
  if (sc.out_exp && blockIdx.x == 0 && threadIdx.x == 0)
    *sc.out_exp = ff3d_out_exp(sc, ff3d_ld_exp(sc.a_exp), false, relu ? upper : INFINITY);
  const bool vec = (N & 3) == 0;
  if (vec) {
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < MN; i += (long long)gridDim.x * 1024) {
      float4 a = *reinterpret_cast<const float4*>(ws + i);
      for (int s = 1; s < S; ++s) {
        const float4 b = *reinterpret_cast<const float4*>(ws + (long long)s * MN + i);
        a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
      }
      const int n = (int)(i % N);
      float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = fmaf(v[k], sc_in, bias ? bias[n + k] : 0.f);
        if (round_bf16) v[k] = (float)(__bf16)v[k];
        if (relu) v[k] = fminf(fmaxf(v[k], 0.f), upper);
      }
      *reinterpret_cast<float4*>(out + i) = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < MN; i += (long long)gridDim.x * 256) {
      float a = ws[i];
      for (int s = 1; s < S; ++s) a += ws[(long long)s * MN + i];
      a = fmaf(a, sc_in, bias ? bias[(int)(i % N)] : 0.f);
      if (round_bf16) a = (float)(__bf16)a;
      if (relu) a = fminf(fmaxf(a, 0.f), upper);
      out[i] = a;
    }
  }
}

// Second half of a split-K CONVOLUTION (round 6, ff3d_conv3x3_f16x3_splitk): the planes hold raw partial sums as (B*Ho*Wo, N) rows
// (pixel-major = NHWC); out is the NCHW fp32 map the reference's tensor boundary wants.  A block reduces a 64-pixel x 64-channel tile
// in slice order (deterministic), applies scale / bias / ReLU and transposes it through LDS: 16-byte reads along channels, 16-byte
// writes along pixels (scalar where a 4-pixel run would cross a frame, HW % 4 != 0).
__global__ __launch_bounds__(256) void splitk_reduce_nchw_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                                 float* __restrict__ out, int M, int N, int HW, int S, int relu,
                                                                 float upper, Ff3dScale sc) {
  __shared__ float tile[64][65];                    // [channel][pixel]
  const float sc_in = ff3d_pow2(ff3d_ld_exp(sc.a_exp) + ff3d_ld_exp(sc.w_exp));
  if (sc.out_exp && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    *sc.out_exp = ff3d_out_exp(sc, ff3d_ld_exp(sc.a_exp), false, relu ? upper : INFINITY);
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int l16 = threadIdx.x & 15, r16 = threadIdx.x >> 4;
  const long long MN = (long long)M * N;
  const bool nvec = (N & 3) == 0;
  for (int r = r16; r < 64; r += 16) {
    const int m = m0 + r, n = n0 + 4 * l16;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (m < M && n < N) {
      const long long o = (long long)m * N + n;
      if (nvec) {
        float4 a = *reinterpret_cast<const float4*>(ws + o);
        for (int q = 1; q < S; ++q) {
          const float4 b = *reinterpret_cast<const float4*>(ws + q * MN + o);
          a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
        }
        v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w;
      } else {
        for (int k = 0; k < 4; ++k)
          if (n + k < N) {
            float a = ws[o + k];
            for (int q = 1; q < S; ++q) a += ws[q * MN + o + k];
            v[k] = a;
          }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = fmaf(v[k], sc_in, (bias && n + k < N) ? bias[n + k] : 0.f);
        if (relu) v[k] = fminf(fmaxf(v[k], 0.f), upper);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[4 * l16 + k][r] = v[k];
  }
  __syncthreads();
  const bool pvec = (HW & 3) == 0;                  // then m0 + 4 * l16 .. + 3 lie in one frame (m0 % 64 == 0)
  for (int c = r16; c < 64; c += 16) {
    const int n = n0 + c, m = m0 + 4 * l16;
    if (n >= N || m >= M) continue;
    if (pvec && m + 3 < M) {
      const int b = m / HW, q = m - b * HW;
      *reinterpret_cast<float4*>(out + ((long long)b * N + n) * HW + q) =
          make_float4(tile[c][4 * l16], tile[c][4 * l16 + 1], tile[c][4 * l16 + 2], tile[c][4 * l16 + 3]);
    } else {
      for (int k = 0; k < 4; ++k)
        if (m + k < M) {
          const int b = (m + k) / HW, q = (m + k) - b * HW;
          out[((long long)b * N + n) * HW + q] = tile[c][4 * l16 + k];
        }
    }
  }
}

// fp32 -> (hi, lo') fp16 split, optionally transposing NCHW -> NHWC (64 pixels x 64 channels per block through LDS).
// Range normalisation (ff3d.h): the planes hold x * 2^-e.  `hint` = {guessed e, max|x| bits, redo flag, -}: the first pass
// converts with the guess while it measures max|x| (one atomicMax per block); split_verify_kernel checks the guess and a
// second, normally empty, pass (redo = 1) re-converts only when the guess was out of range.  hint = 65 * 64 ints: [0..3] +
// 64 maximum slots, 256 bytes apart.
__device__ __forceinline__ void split16(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)((x - (float)hi) * SM_LO_SCALE);
}

// max|x| of the pass: every wave folds its maximum with a shuffle butterfly and issues ONE fire-and-forget atomicMax
// (|x| bit patterns order like unsigned ints) into one of 64 slots that lie 256 bytes apart - different L2 channels, no
// value returned, nothing waits.  (One shared address serialised 65 k atomics in the L2 and doubled the pass; reading the
// running maximum first to skip the atomic put a ~2 us global-load latency at the end of every 1-2 us block.)
constexpr int SPLIT_SLOTS = 64, SPLIT_SLOT_STRIDE = 64;          // ints; slot s lives at hint[(1 + s) * 64]
__device__ __forceinline__ void wave_amax(float m, int* hint, unsigned block_linear) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) {
    const unsigned slot = (block_linear * 4u + (threadIdx.x >> 6)) & (SPLIT_SLOTS - 1);
    __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(hint) + (1 + slot) * SPLIT_SLOT_STRIDE, __float_as_uint(m),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ void split_nchw_to_nhwc_body(const float* __restrict__ x, _Float16* __restrict__ hi,
                                                        _Float16* __restrict__ lo, int C, int HW, int vec4,
                                                        int* __restrict__ hint, int redo, int b, unsigned block_linear,
                                                        int p_tile, int c_tile) {
  __shared__ float tile[64][65];
  if (redo && hint[2] == 0) return;               // second pass: only when the verified exponent differs from the guess
  const float sc = hint ? ff3d_pow2(-hint[0]) : 1.f;
  float amax = 0.f;
  const int p0 = p_tile * 64, c0 = c_tile * 64;
  const float* xb = x + (long long)b * C * HW;
  if (vec4) {   // HW % 4 == 0, C % 4 == 0, 16-byte aligned bases: 16-byte reads along pixels, 8-byte writes along channels
    const int l16 = threadIdx.x & 15, r16 = threadIdx.x >> 4;
    for (int c = r16; c < 64; c += 16) {
      const int cc = c0 + c, pp = p0 + 4 * l16;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cc < C && pp < HW) v = *reinterpret_cast<const float4*>(xb + (long long)cc * HW + pp);
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      tile[c][4 * l16 + 0] = v.x * sc, tile[c][4 * l16 + 1] = v.y * sc, tile[c][4 * l16 + 2] = v.z * sc;
      tile[c][4 * l16 + 3] = v.w * sc;
    }
    __syncthreads();
    for (int q = r16; q < 64; q += 16) {
      const int pp = p0 + q, cc = c0 + 4 * l16;
      if (pp < HW && cc < C) {
        _Float16 h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) split16(tile[4 * l16 + k][q], h[k], l[k]);
        const long long o = ((long long)b * HW + pp) * C + cc;
        *reinterpret_cast<uint2*>(hi + o) = *reinterpret_cast<uint2*>(h);
        *reinterpret_cast<uint2*>(lo + o) = *reinterpret_cast<uint2*>(l);
      }
    }
  } else {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = ty; c < 64; c += 4) {
      const int cc = c0 + c, pp = p0 + tx;
      const float v = (cc < C && pp < HW) ? xb[(long long)cc * HW + pp] : 0.f;
      amax = fmaxf(amax, fabsf(v));
      tile[c][tx] = v * sc;
    }
    __syncthreads();
    for (int q = ty; q < 64; q += 4) {
      const int pp = p0 + q, cc = c0 + tx;
      if (pp < HW && cc < C) {
        _Float16 h, l;
        split16(tile[tx][q], h, l);
        const long long o = ((long long)b * HW + pp) * C + cc;
        hi[o] = h, lo[o] = l;
      }
    }
  }
  if (hint && !redo) wave_amax(amax, hint, block_linear);
}

// Block order of the transposing split (`order`; FF3D_SPLIT_ORDER):
//   0 "pixel"    grid (pixel tiles, channel tiles, frames), pixel tiles fastest - rounds 1-5;
//   1 "channel"  channel tiles fastest: the C / 64 blocks that fill the 128-byte pieces of one pixel's output row run back to back (idea:
//                at 468 x 468 x 256 x 8 frames the planes no longer fit the memory-side cache between the channel passes).  Measured
//                level on all three workloads (profiles/r05_x_split_order_ab.txt);
//   2 "xcd"      1-D grid, XCD-remapped, pixel tiles fastest: neighbouring pixel tiles - which share the 128-byte line their common
//                boundary straddles when a channel plane starts 64 bytes off a line (HW * 4 = 64 mod 128 at 180 x 180 and 468 x 468:
//                PMC FETCH_SIZE 1.30 GB per 32-frame map for 1.06 GB) - run on one XCD: 1.04 GB, the step 1335.0 / 1332.3 vs
//                1331.0 / 1329.8 frames/s same box (profiles/r05_ac_*).  The default.
struct SplitTile {
  int p_tile, c_tile, z;
  unsigned linear;
};
__device__ __forceinline__ SplitTile split_tile(int C, int HW, int order) {
  SplitTile t;
  if (order == 2) {
    const unsigned lid = ff3d_xcd_remap(blockIdx.x, gridDim.x);
    const unsigned pt = (unsigned)((HW + 63) / 64), ct = (unsigned)((C + 63) / 64);
    t.p_tile = (int)(lid % pt), t.c_tile = (int)((lid / pt) % ct), t.z = (int)(lid / (pt * ct)), t.linear = lid;
  } else {
    t.p_tile = order ? blockIdx.y : blockIdx.x, t.c_tile = order ? blockIdx.x : blockIdx.y, t.z = blockIdx.z;
    t.linear = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  }
  return t;
}

__global__ __launch_bounds__(256) void split_nchw_to_nhwc_kernel(const float* __restrict__ x, _Float16* __restrict__ hi,
                                                                 _Float16* __restrict__ lo, int C, int HW, int vec4,
                                                                 int* __restrict__ hint, int redo, int order) {
  const SplitTile t = split_tile(C, HW, order);
  split_nchw_to_nhwc_body(x, hi, lo, C, HW, vec4, hint, redo, t.z, t.linear, t.p_tile, t.c_tile);
}

// Up to four maps of ONE shape in one launch (the stage maps a multi-stage head receives): gridDim.z = members x frames.
constexpr int SPLIT_MAX_GROUP = 4;
struct SplitGroup {
  const float* x[SPLIT_MAX_GROUP];
  _Float16 *hi[SPLIT_MAX_GROUP], *lo[SPLIT_MAX_GROUP];
  int *hint[SPLIT_MAX_GROUP], *out_exp[SPLIT_MAX_GROUP];
  int B;
};
__global__ __launch_bounds__(256) void split_nchw_to_nhwc_group_kernel(SplitGroup gp, int C, int HW, int vec4, int redo, int order) {
  const SplitTile t = split_tile(C, HW, order);
  const int g = t.z / gp.B, b = t.z - g * gp.B;
  split_nchw_to_nhwc_body(gp.x[g], gp.hi[g], gp.lo[g], C, HW, vec4, gp.hint[g], redo, b, t.linear, t.p_tile, t.c_tile);
}

static int split_order(int C, int HW, long long frames) {
  static const int forced = [] {
    const char* e = getenv("FF3D_SPLIT_ORDER");
    return !e ? 2 : (e[0] == 'c' ? 1 : e[0] == 'x' ? 2 : 0);
  }();
  if (forced == 1 && (HW + 63) / 64 > 65535) return 0;
  if (forced == 2 && (long long)((HW + 63) / 64) * ((C + 63) / 64) * frames >= (1ll << 31)) return 0;
  return forced;
}
static dim3 split_grid(int C, int HW, int frames, int order) {
  const unsigned pt = (unsigned)((HW + 63) / 64), ct = (unsigned)((C + 63) / 64);
  return order == 2 ? dim3(pt * ct * (unsigned)frames) : order == 1 ? dim3(ct, pt, frames) : dim3(pt, ct, frames);
}

__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, _Float16* __restrict__ hi,
                                                         _Float16* __restrict__ lo, long long n4, int* __restrict__ hint,
                                                         int redo) {
  if (redo && hint[2] == 0) return;
  const float sc = hint ? ff3d_pow2(-hint[0]) : 1.f;
  float amax = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    _Float16 h[4], l[4];
    split16(v.x * sc, h[0], l[0]);
    split16(v.y * sc, h[1], l[1]);
    split16(v.z * sc, h[2], l[2]);
    split16(v.w * sc, h[3], l[3]);
    reinterpret_cast<uint2*>(hi)[i] = *reinterpret_cast<uint2*>(h);
    reinterpret_cast<uint2*>(lo)[i] = *reinterpret_cast<uint2*>(l);
  }
  if (hint && !redo) wave_amax(amax, hint, blockIdx.x);
}

// One wave: accept the guessed exponent iff 2^5 <= max|x| * 2^-e < 2^15 (no overflow, at most 9 binades of fp16's range
// given away); otherwise take the exponent that puts max|x| into [2^13, 2^14) and flag the redo pass.  Resets the maximum.
__global__ __launch_bounds__(64) void split_verify_kernel(int* __restrict__ hint, int* __restrict__ out_exp) {
  unsigned* slot = reinterpret_cast<unsigned*>(hint) + (1 + threadIdx.x) * SPLIT_SLOT_STRIDE;
  float amax = __uint_as_float(*slot);
  *slot = 0u;                                         // reset for the next conversion
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
  if (out_exp) *out_exp = e_final;
}

__global__ __launch_bounds__(64) void split_verify_group_kernel(SplitGroup gp) {
  int* hint = gp.hint[blockIdx.x];
  int* out_exp = gp.out_exp[blockIdx.x];
  unsigned* slot = reinterpret_cast<unsigned*>(hint) + (1 + threadIdx.x) * SPLIT_SLOT_STRIDE;
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
  if (out_exp) *out_exp = e_final;
}

template <int WM, int NBUF, bool TR, int PL = 2, bool CONVB = false>
int launch_variant(const SplitMMParams& p, hipStream_t s) {
  constexpr int BM = WM * 64;
  constexpr size_t lds_bytes = (size_t)NBUF * PL * (BM + SM_BN) * SM_BK * sizeof(_Float16);
  static bool configured[64] = {};                // > 64 KiB of dynamic LDS has to be enabled once per kernel AND device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!configured[dev & 63]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&splitmm_kernel<WM, NBUF, TR, PL, CONVB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
      return FF3D_ERR_LAUNCH;
    configured[dev & 63] = true;
  }
  const int m_tiles = p.period ? p.nbatch * ((p.period + BM - 1) / BM) : (p.M + BM - 1) / BM;
  const int blocks = m_tiles * ((p.N + SM_BN - 1) / SM_BN);
  ff3d_clear_error();
  hipLaunchKernelGGL((splitmm_kernel<WM, NBUF, TR, PL, CONVB>), dim3(blocks, p.ksplit > 1 ? p.ksplit : 1), dim3(WM * 128), lds_bytes, s, p);
  return ff3d_launch_status();
}


// ---------------------------------------------------------------------------------------------------------------------
// Weight-stationary GEMM for short K (K <= 256: value_proj of the decoder, M = B * Nv ~ 1.4 M rows, K = C, N = layers * C).
// In splitmm_kernel every 128 x 128 tile re-streams its 128 x K weight block through the LDS DMA: for the value GEMM that is
// 8.2 GB of weight pieces per launch on top of the 8.4 GB of activations (6 N-tiles each read A once), all of it L2 -> LDS
// traffic, for 8 K-steps of MFMAs between a pipeline fill and a 64 KiB epilogue.  Here a block owns ONE N-tile for a long run
// of M-tiles and keeps its weight fragments IN REGISTERS: 4 waves (one per SIMD, the unified 512-register file), wave tile
// 128 rows x 32 columns, B fragments 2 (j) x K/32 (steps) x 2 planes x 4 = 128 VGPRs at K = 256, accumulators 8 x 2 tiles x 2
// x 4 = 128 (AGPRs).  Only the A tile moves: per TWO K-steps a 128-row x 64-k slot (full 128-byte lines per row and plane,
// 8 DMA pieces per wave; splitmm_kernel issues 16 half-line pieces for the same MFMA work) into a 4-deep LDS ring (128 KiB),
// issued 3 slots ahead (96 KiB in flight per CU); every wave reads the whole A slot (32 fragment reads for 96 MFMAs, 128 KiB
// per slot and CU = 1024 LDS cycles under 1536 MFMA cycles), one raw s_barrier per slot, the next slot's first fragments
// fetched under this slot's MFMAs.  The K loop never drains at a tile boundary: the ring runs across tiles, and the epilogue's stores retire behind
// the next tile's steps - loads and stores share the in-order VM counter, so the waits count both.
// (Tried: 256 columns per block - 8 waves, two per SIMD, the two accumulators folded into one with unscaled low parts - to halve
// the A stream: at K = 128, N = 768 1.39 vs 1.53 ms, at K = 256 the 256-register budget spills (3.3 vs 2.1 ms); what is left is
// mostly the 4.18 GB of fp32 output: the stores alone take 1.0 ms.  Non-temporal stores for it: 2.35 vs 2.12 ms.)
// s_waitcnt vmcnt(n) for a run-time n (a multiple of 4 up to 60; the instruction takes an immediate)
__device__ __forceinline__ void ws_wait_vm(int n) {
  switch (n) {
#define FF3D_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    FF3D_VM(4) FF3D_VM(8) FF3D_VM(12) FF3D_VM(16) FF3D_VM(20) FF3D_VM(24) FF3D_VM(28) FF3D_VM(32) FF3D_VM(36) FF3D_VM(40)
    FF3D_VM(44) FF3D_VM(48) FF3D_VM(52) FF3D_VM(56) FF3D_VM(60)
#undef FF3D_VM
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// Round 3: NJ = 16-column tiles per wave.  NJ = 2 is the kernel of round 2 (block = 128 columns).  NJ = 3 (block = 192 columns,
// N = 768 -> 4 column tiles instead of 6; weights 192 + accumulators 192 registers) would amortise the A stream over 1.5 x the
// MFMAs, but at K = 256 it spills and is 2.4 x SLOWER (5.0 vs 2.07 ms) - kept as an opt-in instance, see launch_ws.
// ABL: timing ablations (tuning only, WRONG results): 1 no MFMA, 2 no DMA, 4 no fragment reads, 8 no stores
// PER: the periodic form (ff3d_gemm_f16x3_rowbias): M = nbatch frames of `period` rows, out += bias_tab[row within the frame].
// Tiles never straddle frames and a block walks them FRAME-FASTEST: the table tile of a row block (128 x 64 NJ fp32) is loaded
// once into registers (scaled by the inverse operand scale, exact) and enters every frame's accumulators as their initial
// value - 64 registers that this one-wave-per-SIMD kernel has to spare.
// PL (round 5) = operand planes: 1 = the bf16 instance (one plane per operand on v_mfma_f32_16x16x32_bf16, see splitmm_kernel):
// value_proj of BASELINE configs[4] - A = the bf16 (pyramid + pos-embed) plane bev_flatten writes, weights bf16 in registers (half the
// registers: NJ = 4 column tiles fit where the split form spills), result rounded once to bf16 and stored as bf16 rows (out_mode 3:
// what the deformable gather reads, half the store bytes of the fp32 form) or as fp32 (out_mode 0).
// WV (round 5) = waves per block: 4 (one per SIMD, the form above) or 8 with NJ = 1 - the same 128 columns per block as 4 x NJ = 2, but
// every wave holds half the weights and accumulators (~200 registers), so TWO waves share a SIMD and one's MFMAs cover the other's
// LDS waits; the price is that 8 waves read the A slot instead of 4 (LDS fragment traffic x 2).
template <int KS, int NJ, int ABL = 0, bool PER = false, int PL = 2, int WV = 4>
__global__ __launch_bounds__(64 * WV, WV == 8 ? 2 : 1) void splitmm_ws_kernel(SplitMMParams p, int groups) {
  // Ring of NB slots, one slot = the A tile of TWO K-steps (128 rows x 64 k: full 128-byte lines per row and plane, 32 KiB),
  // DMA issued PD slots ahead; at iteration t the barrier makes slot t+1 visible (one early: the first fragments of the next
  // slot are fetched under this slot's MFMAs).
  constexpr int T = 64 * WV, BM = 128, NB = 4, PD = 3, RK = 2 * SM_BK, RS = KS / 2;  // RS ring steps per tile
  constexpr int RP = T / 8, NQ = BM / RP;                                            // rows a staging pass covers; passes per plane
  constexpr int A_PLANE = BM * RK, BUF = PL * A_PLANE, PIECES = NQ * PL;             // halves; DMA instructions per thread and slot
  constexpr int WN = 16 * NJ, BN = WV * WN, ST = 8 * NJ;                             // wave / block columns; stores per wave and tile
  static_assert(KS % 2 == 0, "K must be a multiple of 64");
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];                     // [NB][A_hi | A_lo] + bias tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int rb_frame = PER ? (p.period + BM - 1) / BM : 0;             // row blocks per frame
  const int n_tiles = (p.N + BN - 1) / BN, m_tiles = PER ? p.nbatch * rb_frame : (p.M + BM - 1) / BM;
  const unsigned lid = ff3d_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = (int)(lid % n_tiles), g = (int)(lid / n_tiles);       // the n_tiles blocks of a group walk the same M-tiles
  const int per = (m_tiles + groups - 1) / groups;
  const int t_lo = g * per, t_hi = min(m_tiles, t_lo + per);
  if (t_lo >= t_hi) return;
  const int n0 = nt * BN, nw = n0 + wave * WN;                         // this wave's WN output columns

  // ---- weight fragments -> registers (once per block); bias tile -> LDS
  half8 bh[NJ][KS], bl[NJ][KS];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = nw + j * 16 + fr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const unsigned o = n < p.N ? ((unsigned)n * (unsigned)p.K + (unsigned)(ks * SM_BK + kq * 8)) * 2u : p.b_zero;
      bh[j][ks] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(p.w_hi) + o);
      if (PL == 2) bl[j][ks] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(p.w_lo) + o);
    }
  }
  float* const s_bias = reinterpret_cast<float*>(lds + NB * BUF);      // this N-tile's BN bias values (0 beyond N)
  if (tid < BN) s_bias[tid] = (p.bias && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
  const float sc_in = ff3d_pow2(ff3d_ld_exp(p.sc.a_exp) + ff3d_ld_exp(p.sc.w_exp));
  float sc_pair = 1.f;                              // pair output: 2^-e_out of the layer's bound exponent
  const float sc_res = p.res_hi ? ff3d_pow2(ff3d_ld_exp(p.sc.res_exp)) : 1.f;
  if (p.sc.out_exp) {
    const int e_out = ff3d_out_exp(p.sc, ff3d_ld_exp(p.sc.a_exp), p.res_hi != nullptr, p.relu ? p.upper : INFINITY);
    if (p.out_mode == 2) sc_pair = ff3d_pow2(-e_out);
    if (lid == 0 && tid == 0) *p.sc.out_exp = e_out;
  }

  // ---- A staging.  LDS rows are 128 B (8 chunks of 16 B); the chunk index is XOR-swizzled with h(row) = (row >> 1) & 7 - on
  // the DMA's SOURCE address (its destination is lane-linear) and again on the fragment read: a ds_read_b128 service group
  // (lanes {0-3, 12-15, 20-27}, ...) then touches 16 distinct 16-byte bank columns, column = (row & 1) * 8 + (chunk ^ h).
  // Thread owns slots q*T + tid, q < NQ, of each plane: row (tid >> 3) + RP q, chunk tid & 7 (h is the same for all q: RP % 16 == 0).
  const int a_row0 = tid >> 3;
  const unsigned a_sw = (unsigned)(((tid & 7) ^ ((a_row0 >> 1) & 7)) * 16);
  auto stage = [&](int rstep) {                   // rstep = (tile - t_lo) * RS + rs, into ring slot rstep % NB
    if (ABL & 2) return;
    const int tile = t_lo + rstep / RS, rs = rstep - (rstep / RS) * RS;
    _Float16* base = lds + (rstep & (NB - 1)) * BUF;
    int tm0 = tile * BM, tm_end = p.M;             // first row of the tile, end of the rows it may read
    if (PER) {
      const int rb = tile / p.nbatch, f = tile - rb * p.nbatch;
      tm0 = f * p.period + rb * BM, tm_end = (f + 1) * p.period;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int m = tm0 + a_row0 + RP * q;
      const unsigned ao = (m < tm_end ? (unsigned)m * (unsigned)p.K * 2u + (unsigned)(rs * (RK * 2)) : p.a_zero) + a_sw;
      _Float16* dst = base + (q * T + wave * 64) * 8;
      glds16(p.a_hi, ao, dst);
      if (PL == 2) glds16(p.a_lo, ao, dst + A_PLANE);
    }
  };
  // fragment of M-tile i, K-substep sub: row i*16 + fr, chunk (sub*4 + kq) ^ h(row); h depends on fr only (16 % 16 == 0)
  const int a_h = (fr >> 1) & 7;
  const int a_rd0 = fr * RK + ((kq ^ a_h) * 8), a_rd1 = fr * RK + (((4 + kq) ^ a_h) * 8);

  const int steps = (t_hi - t_lo) * RS;
  __builtin_amdgcn_s_waitcnt(0);                  // the weight loads are out of the counted window
  __syncthreads();
  for (int q = 0; q < PD; ++q)
    if (q < steps) stage(q);
  // slot 0 has to be visible before the loop (inside it the barrier of iteration t publishes slot t + 1)
  ws_wait_vm(min(PD - 1, steps - 1) * PIECES);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // Fragment pipeline (round 3).  The ISA of rounds 1-2 showed the compiler sinking every fragment read to just before its
  // first MFMA (one register quad reused for all 16 groups of a slot: ds_read_b128, s_waitcnt lgkmcnt(0), 6 MFMAs, ...), i.e. a
  // full LDS round trip exposed per 96 cycles of MFMA work with one wave per SIMD - the "56 % MFMA-busy whatever the tile"
  // plateau.  Now: group g (= rs * 16 + u inside a tile) uses fh / fl[g % 3], the reads of group g + 2 are issued BEFORE group
  // g's MFMAs and __builtin_amdgcn_sched_barrier(0) pins that order; the last two groups of a slot fetch the first two of the
  // next (already published) slot.  The rotation phase is static inside a tile (the loops are fully unrolled) and is re-based
  // with register moves at the tile end.
  half8 fh[3], fl[3];
  fh[0] = *reinterpret_cast<const half8*>(lds + a_rd0);
  fh[1] = *reinterpret_cast<const half8*>(lds + a_rd0 + 16 * RK);
  if (PL == 2) {
    fl[0] = *reinterpret_cast<const half8*>(lds + A_PLANE + a_rd0);
    fl[1] = *reinterpret_cast<const half8*>(lds + A_PLANE + a_rd0 + 16 * RK);
  } else {
    fl[0] = fh[0], fl[1] = fh[1];                 // (unused in the one-plane instance)
  }
  fh[2] = fh[0], fl[2] = fl[0];

  // The in-order VM counter also counts the epilogues' stores (ST per wave on a full tile).  Pieces of slot s are issued at
  // iteration s - PD; the stores of the epilogue after iteration E are issued behind stage(E + PD), so they are younger than
  // every slot <= E + PD: while the slot being waited for is one of those, the ST stores stay in the allowance.
  int e_last = -1000, e_prev = -1000;             // iterations of the last two epilogues with countable stores
  int step = 0;
  f32x4 tabv[PER ? 8 : 1][PER ? NJ : 1];          // PER: the row block's table tile / operand scale, in accumulator layout
  int tab_rb = -1;
  const float sc_inv = 1.f / sc_in;
  for (int tile = t_lo; tile < t_hi; ++tile) {
    if (PER) {
      const int rb = tile / p.nbatch;
      if (rb != tab_rb) {                          // (once per nbatch tiles; the compiler's wait for these loads drains the ring)
        tab_rb = rb;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = rb * BM + i * 16 + fr;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int n = nw + j * 16 + kq * 4;
            float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < p.period && n + 3 < p.N && (p.N & 3) == 0) {
              t4 = *reinterpret_cast<const float4*>(p.bias_tab + (long long)row * p.N + n);
            } else if (row < p.period) {
              const float* tp = p.bias_tab + (long long)row * p.N;
              t4 = make_float4(n < p.N ? tp[n] : 0.f, n + 1 < p.N ? tp[n + 1] : 0.f, n + 2 < p.N ? tp[n + 2] : 0.f,
                               n + 3 < p.N ? tp[n + 3] : 0.f);
            }
            tabv[i][j] = f32x4{t4.x * sc_inv, t4.y * sc_inv, t4.z * sc_inv, t4.w * sc_inv};
          }
        }
      }
    }
    f32x4 acc_m[8][NJ], acc_x[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc_m[i][j] = PER ? tabv[i][j] : f32x4{0.f, 0.f, 0.f, 0.f}, acc_x[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rs = 0; rs < RS; ++rs, ++step) {
      const int need = step + 1;                  // publish slot step + 1 (its first fragments are fetched in this iteration)
      if (ABL & 10) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (need < steps) {
        const int younger = min(step + PD - 1, steps - 1) - need;       // issued slots behind `need`
        const int stores = (need <= e_last + PD ? ST : 0) + (need <= e_prev + PD ? ST : 0);
        ws_wait_vm(min(younger * PIECES + stores, 60));                 // (a smaller allowance is only stricter)
      }
      __builtin_amdgcn_s_barrier();               // ... for every wave; all reads of slot step - 1 (and step's first) are retired
      asm volatile("" ::: "memory");
      if (step + PD < steps) stage(step + PD);    // ring slot (step + PD) % NB = the one slot step - 1 used
      const _Float16* t = lds + (step & (NB - 1)) * BUF;
      const _Float16* tn = lds + (need & (NB - 1)) * BUF;
#pragma unroll
      for (int u = 0; u < 16; ++u) {              // u = sub * 8 + i: the two K-substeps of the slot, 8 M-tiles each
        const int sub = u >> 3, i = u & 7, ks = rs * 2 + sub;
        const int cur = (rs * 16 + u) % 3, nxt = (rs * 16 + u + 2) % 3;
        if (!(ABL & 4)) {
          if (u < 14) {
            const int un = u + 2, off = ((un >> 3) ? a_rd1 : a_rd0) + (un & 7) * 16 * RK;
            fh[nxt] = *reinterpret_cast<const half8*>(t + off);
            if (PL == 2) fl[nxt] = *reinterpret_cast<const half8*>(t + A_PLANE + off);
          } else if (need < steps) {              // groups 0 / 1 of the next slot (already published)
            const int off = a_rd0 + (u - 14) * 16 * RK;
            fh[nxt] = *reinterpret_cast<const half8*>(tn + off);
            if (PL == 2) fl[nxt] = *reinterpret_cast<const half8*>(tn + A_PLANE + off);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        const half8 ah = fh[cur], al = fl[cur];
        if (ABL & 1) {
          asm volatile("" ::"v"(ah), "v"(al));
        } else if (PL == 1) {
          const bf16x8 ab = __builtin_bit_cast(bf16x8, ah);
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc_m[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bh[j][ks]), ab, acc_m[i][j], 0, 0, 0);
        } else {
          // transposed accumulators: a lane holds 4 consecutive columns of one row.  Pass-major over the wave's column tiles:
          // the dependent acc_x MFMAs of a tile are NJ instructions apart
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc_m[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j][ks], ah, acc_m[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc_x[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j][ks], ah, acc_x[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc_x[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j][ks], al, acc_x[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {   // re-base the rotation: the next tile's groups 0 / 1 sit in f[(RS * 16) % 3], f[(RS * 16 + 1) % 3]
      const half8 h0 = fh[(RS * 16) % 3], l0 = fl[(RS * 16) % 3], h1 = fh[(RS * 16 + 1) % 3], l1 = fl[(RS * 16 + 1) % 3];
      fh[0] = h0, fl[0] = l0, fh[1] = h1, fl[1] = l1;
    }
    // ---- epilogue of this tile: fp32 row-major; on a full tile exactly ST store instructions per wave (16 B per lane)
    int m0 = tile * BM, m_end = p.M;
    if (PER) {
      const int rb = tile / p.nbatch, f = tile - rb * p.nbatch;
      m0 = f * p.period + rb * BM, m_end = (f + 1) * p.period;
    }
    const bool full = m0 + BM <= m_end && n0 + BN <= p.N && (p.N & 3) == 0;
    float bv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float4 bj = *reinterpret_cast<const float4*>(s_bias + wave * WN + j * 16 + kq * 4);
      bv[j][0] = bj.x, bv[j][1] = bj.y, bv[j][2] = bj.z, bv[j][3] = bj.w;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {                 // j inner: the 64-byte pieces of a row's run back to back
      const int m = m0 + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = nw + j * 16 + kq * 4;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(acc_m[i][j][r] + acc_x[i][j][r] * SM_LO_INV, sc_in, bv[j][r]);
        if (PL == 2 && p.res_hi) {
          // residual pair (M, N) added before the activation (round 5: the 2C -> C mixes of the fusion neck on this kernel).  These
          // loads are younger than every DMA in flight: the wait the compiler puts in front of their use drains the ring once per
          // tile, and the hand-counted waits stay valid (unaccounted YOUNGER operations only make them stricter)
          const long long o2 = (long long)m * p.N + n;
          if (full) {
            const uint2 rh = *reinterpret_cast<const uint2*>(p.res_hi + o2), rl = *reinterpret_cast<const uint2*>(p.res_lo + o2);
            const _Float16* h4 = reinterpret_cast<const _Float16*>(&rh);
            const _Float16* l4 = reinterpret_cast<const _Float16*>(&rl);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf((float)h4[r] + (float)l4[r] * SM_LO_INV, sc_res, v[r]);
          } else if (m < m_end) {
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) v[r] = fmaf((float)p.res_hi[o2 + r] + (float)p.res_lo[o2 + r] * SM_LO_INV, sc_res, v[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (PL == 1) v[r] = (float)(__bf16)v[r];
          if (p.relu) v[r] = fminf(fmaxf(v[r], 0.f), p.upper);
        }
        if (PL == 2 && p.out_mode == 2) {
          // (hi, lo') pair rows (round 5: the 1x1-conv layers of the fusion neck with a pair output, K = 128 / 256): the exponent of
          // the layer's guaranteed bound as in splitmm_kernel; 4 consecutive channels = one 8-byte store per plane
          _Float16 h[4], l[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float vs = v[r] * sc_pair;
            h[r] = (_Float16)vs;
            l[r] = (_Float16)((vs - (float)h[r]) * SM_LO_SCALE);
          }
          const long long o2 = (long long)m * p.N + n;
          if (full) {
            *reinterpret_cast<uint2*>(p.out_hi + o2) = *reinterpret_cast<uint2*>(h);
            *reinterpret_cast<uint2*>(p.out_lo + o2) = *reinterpret_cast<uint2*>(l);
          } else if (m < m_end) {
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) p.out_hi[o2 + r] = h[r], p.out_lo[o2 + r] = l[r];
          }
          continue;
        }
        if (PL == 1 && p.out_mode == 3) {
          // bf16 rows: one 8-byte store per lane and tile (same instruction count as the fp32 form).  Measured (round 5, session e):
          // pairing two column tiles per lane through a shuffle so that every store is 16 bytes (64 contiguous bytes per row and
          // instruction instead of 32) is SLOWER, 2.15 vs 1.69 ms at 2.3 M rows - as the 16-byte transposed stores were for the
          // fp32 tile-streaming kernel (launch()): an instruction then touches 16 rows x 64 B instead of 16 rows x 32 B twice
          __bf16 q[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          __bf16* ob = reinterpret_cast<__bf16*>(p.out_hi) + (long long)m * p.N + n;
          if (full) {
            *reinterpret_cast<uint2*>(ob) = *reinterpret_cast<uint2*>(q);
          } else if (m < m_end) {
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) ob[r] = q[r];
          }
          continue;
        }
        float* o = p.out + (long long)m * p.N + n;
        if (ABL & 8) {
          if (v[0] == 1.2345e-30f) *o = v[1] + v[2] + v[3];        // keeps the arithmetic alive, never stores
        } else if (full) {
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else if (m < m_end) {
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) o[r] = v[r];
        }
      }
    }
    if (full) {
      e_prev = e_last, e_last = step - 1;
    } else {                                      // ragged tile: an unknown number of stores - drain everything
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      e_prev = e_last = -1000;
    }
  }
}

// Round 6 (measured, NOT the default; compiled with FF3D_BUILD_EXPERIMENTS=1, selected with FF3D_GEMM_WS1=1): the weight-stationary GEMM
// with 256 columns per block (value_proj: N = 768 -> 3 column tiles instead of 6).  Reasoning: per ring slot the 128-column form moves
// 32 KB of DMA writes + 4 waves x 32 KB of fragment reads = 160 KB through the LDS (1250 cycles at 128 B / cycle) under 1536 cycles of
// MFMAs, and every A tile is pulled L2 -> LDS by SIX blocks; both ratios depend on the columns a wave owns (A bytes per flop = 1 / wave
// columns), and this kernel doubles them: 4 waves x 64 columns, weight fragments 2 planes x 4 x 8 x 4 = 256 registers, 64-row tiles
// (2 accumulator sets x 4 x 4 x 4 = 128 registers; 512 allocated, 6 spilled outside the K loop), ring of 8 slots x 16 KB issued 7
// ahead; results bit-identical to the 128-column form.  Measured at M = 1 360 800, K = 256, N = 768 (profiles/r06_g_value_gemm_ws1_ab.txt):
//   * this two-accumulator form: 2.25 - 2.28 vs 2.15 - 2.17 ms - SLOWER;
//   * a ONE-accumulator variant (low parts unscaled in the kernel, all three passes into one fp32 accumulator, 422 registers): 2.01 -
//     2.02 vs 2.14 - 2.16 ms alone, 1.85 - 1.90 vs 2.05 ms inside the step, the step 1270 / 1276 vs 1258 / 1260 frames/s (+1.1 %) - but
//     it loses the low part of rows more than 2^16 below the operand's maximum (fp16 subnormals): 1.6e-5 of such a row's own scale
//     against fp64, outside the fp32-class contract of the dense kernels - not shipped for 1 %;
//   * the ablations of this form (same file): stores alone 0.70 ms, loads alone 0.90, MFMA + fragment reads 1.29, MFMA + DMA 0.94,
//     everything but the stores 1.55 - the 0.70 ms of stores add in full.  Loads and stores share the in-order VM counter of the one
//     wave a SIMD holds: a slot issued after a tile's stores cannot be consumed before those stores have retired, and at the rate the
//     MFMAs produce tiles (4.18 GB per ~0.9 ms) the write path runs at the HBM roof, where a store takes longer to retire than the
//     ring (all the LDS there is) can cover.  A separate loader wave would need a second register allocation in the workgroup.  The
//     value GEMM stays at 2.0 - 2.1 ms on the 128-column form; fifteen measured variants over five rounds.
// ABL (FF3D_WS_ABLATE): timing ablations as for the kernel above - 1 no MFMA, 2 no DMA, 4 no fragment reads, 8 no stores.
#ifdef FF3D_BUILD_EXPERIMENTS
template <int KS, int ABL = 0>
__global__ __launch_bounds__(256, 1) void splitmm_ws1_kernel(SplitMMParams p, int groups) {
  constexpr int T = 256, BM = 64, NB = 8, PD = 7, RK = 2 * SM_BK, RS = KS / 2, NJ = 4, MI = BM / 16, GS = 2 * MI;   // GS groups per slot
  constexpr int RP = T / 8, NQ = BM / RP;                                            // rows a staging pass covers; passes per plane
  constexpr int A_PLANE = BM * RK, BUF = 2 * A_PLANE, PIECES = NQ * 2;               // halves; DMA instructions per thread and slot
  constexpr int WN = 16 * NJ, BN = 4 * WN, ST = MI * NJ;                             // wave / block columns; stores per wave and tile
  static_assert(KS % 2 == 0 && NB == PD + 1 && (PD - 1) * PIECES + 2 * ST <= 60, "ring / wait-count arithmetic");
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];                     // [NB][A_hi | A_lo] + bias tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kq = lane >> 4;
  const int n_tiles = (p.N + BN - 1) / BN, m_tiles = (p.M + BM - 1) / BM;
  const unsigned lid = ff3d_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = (int)(lid % n_tiles), g = (int)(lid / n_tiles);       // the n_tiles blocks of a group walk the same M-tiles
  const int per = (m_tiles + groups - 1) / groups;
  const int t_lo = g * per, t_hi = min(m_tiles, t_lo + per);
  if (t_lo >= t_hi) return;
  const int n0 = nt * BN, nw = n0 + wave * WN;

  // ---- weight fragments -> registers (once per block); bias tile -> LDS
  half8 bh[NJ][KS], bl[NJ][KS];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = nw + j * 16 + fr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const unsigned o = n < p.N ? ((unsigned)n * (unsigned)p.K + (unsigned)(ks * SM_BK + kq * 8)) * 2u : p.b_zero;
      bh[j][ks] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(p.w_hi) + o);
      bl[j][ks] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(p.w_lo) + o);
    }
  }
  float* const s_bias = reinterpret_cast<float*>(lds + NB * BUF);
  if (tid < BN) s_bias[tid] = (p.bias && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
  const float sc_in = ff3d_pow2(ff3d_ld_exp(p.sc.a_exp) + ff3d_ld_exp(p.sc.w_exp));

  // ---- A staging: LDS rows of 128 B, chunk index XOR-swizzled with h(row) = (row >> 1) & 7 on the DMA's source address and on the
  // fragment read (see splitmm_ws_kernel).  Thread owns row (tid >> 3) + RP q, chunk tid & 7 of each plane.
  const int a_row0 = tid >> 3;
  const unsigned a_sw = (unsigned)(((tid & 7) ^ ((a_row0 >> 1) & 7)) * 16);
  auto stage = [&](int rstep) {                   // rstep = (tile - t_lo) * RS + rs, into ring slot rstep % NB
    if (ABL & 2) return;
    const int tile = t_lo + rstep / RS, rs = rstep - (rstep / RS) * RS;
    _Float16* base = lds + (rstep & (NB - 1)) * BUF;
    const int tm0 = tile * BM;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int m = tm0 + a_row0 + RP * q;
      const unsigned ao = (m < p.M ? (unsigned)m * (unsigned)p.K * 2u + (unsigned)(rs * (RK * 2)) : p.a_zero) + a_sw;
      _Float16* dst = base + (q * T + wave * 64) * 8;
      glds16(p.a_hi, ao, dst);
      glds16(p.a_lo, ao, dst + A_PLANE);
    }
  };
  const int a_h = (fr >> 1) & 7;
  const int a_rd0 = fr * RK + ((kq ^ a_h) * 8), a_rd1 = fr * RK + (((4 + kq) ^ a_h) * 8);

  const int steps = (t_hi - t_lo) * RS;
  __builtin_amdgcn_s_waitcnt(0);                  // the weight loads are out of the counted window
  __syncthreads();
  for (int q = 0; q < PD; ++q)
    if (q < steps) stage(q);
  ws_wait_vm(min(PD - 1, steps - 1) * PIECES);    // slot 0 visible before the loop (the barrier of iteration t publishes slot t + 1)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // fragment pipeline: group g uses fh / fl[g % 3], the reads of group g + 2 are issued before group g's MFMAs
  half8 fh[3], fl[3];
  fh[0] = *reinterpret_cast<const half8*>(lds + a_rd0);
  fh[1] = *reinterpret_cast<const half8*>(lds + a_rd0 + 16 * RK);
  fl[0] = *reinterpret_cast<const half8*>(lds + A_PLANE + a_rd0);
  fl[1] = *reinterpret_cast<const half8*>(lds + A_PLANE + a_rd0 + 16 * RK);
  fh[2] = fh[0], fl[2] = fl[0];

  int e_last = -1000, e_prev = -1000;             // iterations of the last two epilogues with countable stores
  int step = 0;
  for (int tile = t_lo; tile < t_hi; ++tile) {
    f32x4 acc[MI][NJ], accx[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}, accx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rs = 0; rs < RS; ++rs, ++step) {
      const int need = step + 1;                  // publish slot step + 1 (its first fragments are fetched in this iteration)
      if (ABL & 10) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (need < steps) {
        const int younger = min(step + PD - 1, steps - 1) - need;       // issued slots behind `need`
        const int stores = (need <= e_last + PD ? ST : 0) + (need <= e_prev + PD ? ST : 0);
        ws_wait_vm(min(younger * PIECES + stores, 60));
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (step + PD < steps) stage(step + PD);    // ring slot (step + PD) % NB = the one slot step - 1 used
      const _Float16* t = lds + (step & (NB - 1)) * BUF;
      const _Float16* tn = lds + (need & (NB - 1)) * BUF;
#pragma unroll
      for (int u = 0; u < GS; ++u) {              // u = sub * MI + i: the two K-substeps of the slot, MI row tiles each
        const int sub = u / MI, i = u % MI, ks = rs * 2 + sub;
        const int cur = (rs * GS + u) % 3, nxt = (rs * GS + u + 2) % 3;
        if (ABL & 4) {
        } else if (u < GS - 2) {
          const int un = u + 2, off = ((un / MI) ? a_rd1 : a_rd0) + (un % MI) * 16 * RK;
          fh[nxt] = *reinterpret_cast<const half8*>(t + off);
          fl[nxt] = *reinterpret_cast<const half8*>(t + A_PLANE + off);
        } else if (need < steps) {                // groups 0 / 1 of the next slot (already published)
          const int off = a_rd0 + (u - (GS - 2)) * 16 * RK;
          fh[nxt] = *reinterpret_cast<const half8*>(tn + off);
          fl[nxt] = *reinterpret_cast<const half8*>(tn + A_PLANE + off);
        }
        __builtin_amdgcn_sched_barrier(0);
        const half8 ah = fh[cur], al = fl[cur];
        if (ABL & 1) {
          asm volatile("" ::"v"(ah), "v"(al));
        } else {
          // transposed accumulators (a lane holds 4 consecutive columns of one row); pass-major over the 4 column tiles: the
          // dependent MFMAs of one accumulator are 4 instructions apart
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j][ks], ah, acc[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < NJ; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j][ks], ah, accx[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < NJ; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j][ks], al, accx[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {   // re-base the rotation: the next tile's groups 0 / 1 sit in f[(RS * GS) % 3], f[(RS * GS + 1) % 3]
      const half8 h0 = fh[(RS * GS) % 3], l0 = fl[(RS * GS) % 3], h1 = fh[(RS * GS + 1) % 3], l1 = fl[(RS * GS + 1) % 3];
      fh[0] = h0, fl[0] = l0, fh[1] = h1, fl[1] = l1;
    }
    // ---- epilogue: fp32 row-major; on a full tile exactly ST 16-byte store instructions per wave
    const int m0 = tile * BM;
    const bool full = m0 + BM <= p.M && n0 + BN <= p.N && (p.N & 3) == 0;
    float bv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float4 bj = *reinterpret_cast<const float4*>(s_bias + wave * WN + j * 16 + kq * 4);
      bv[j][0] = bj.x, bv[j][1] = bj.y, bv[j][2] = bj.z, bv[j][3] = bj.w;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {                // j inner: the 64-byte pieces of a row's 256-byte run back to back
      const int m = m0 + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = nw + j * 16 + kq * 4;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fmaf(acc[i][j][r] + accx[i][j][r] * SM_LO_INV, sc_in, bv[j][r]);
          if (p.relu) v[r] = fminf(fmaxf(v[r], 0.f), p.upper);
        }
        float* o = p.out + (long long)m * p.N + n;
        if (ABL & 8) {
          if (v[0] == 1.2345e-30f) *o = v[1] + v[2] + v[3];        // keeps the arithmetic alive, never stores
        } else if (full) {
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else if (m < p.M) {
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) o[r] = v[r];
        }
      }
    }
    if (full) {
      e_prev = e_last, e_last = step - 1;
    } else {                                      // ragged tile: an unknown number of stores - drain everything
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      e_prev = e_last = -1000;
    }
  }
}

int launch_ws1(const SplitMMParams& p, hipStream_t s) {
  constexpr int BN = 256, BM = 64;
  static int cus[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!cus[dev & 63]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return FF3D_ERR_LAUNCH;
    cus[dev & 63] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int n_tiles = (p.N + BN - 1) / BN, m_tiles = (p.M + BM - 1) / BM;
  int groups = cus[dev & 63] / n_tiles;
  if (groups < 1) groups = 1;
  if (groups > m_tiles) groups = m_tiles;
  const dim3 grid((unsigned)(groups * n_tiles)), block(256);
  constexpr size_t lds_bytes = 8 * 2 * BM * 2 * SM_BK * sizeof(_Float16) + BN * sizeof(float);     // 128 KiB ring + bias tile
  ff3d_clear_error();
#define FF3D_WS1(KS)                                                                                                       \
  do {                                                                                                                     \
    static bool configured[64] = {};                                                                                       \
    if (!configured[dev & 63]) {                                                                                           \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&splitmm_ws1_kernel<KS>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds_bytes) != hipSuccess)                                                               \
        return FF3D_ERR_LAUNCH;                                                                                            \
      configured[dev & 63] = true;                                                                                         \
    }                                                                                                                      \
    hipLaunchKernelGGL((splitmm_ws1_kernel<KS>), grid, block, lds_bytes, s, p, groups);                                    \
  } while (0)
#ifdef FF3D_BUILD_EXPERIMENTS
  static const int abl = [] {                     // timing ablations (tuning only, WRONG results): FF3D_WS_ABLATE = bit mask
    const char* e = getenv("FF3D_WS_ABLATE");
    return e ? atoi(e) : 0;
  }();
  if (abl && p.K == 256) {
#define FF3D_WS1A(n)                                                                                                       \
  case n:                                                                                                                  \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&splitmm_ws1_kernel<8, n>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds_bytes);                                                                             \
    hipLaunchKernelGGL((splitmm_ws1_kernel<8, n>), grid, block, lds_bytes, s, p, groups);                                  \
    break;
    switch (abl) { FF3D_WS1A(1) FF3D_WS1A(2) FF3D_WS1A(4) FF3D_WS1A(8) FF3D_WS1A(6) FF3D_WS1A(7) FF3D_WS1A(9) FF3D_WS1A(10) FF3D_WS1A(12) default: break; }
#undef FF3D_WS1A
    return ff3d_launch_status();
  }
#endif
  if (p.K == 256)
    FF3D_WS1(8);
  else
    return FF3D_ERR_UNSUPPORTED;
#undef FF3D_WS1
  return ff3d_launch_status();
}

#endif  // FF3D_BUILD_EXPERIMENTS (splitmm_ws1_kernel)

// Grid of the weight-stationary form: n_tiles * groups blocks, groups = CUs / n_tiles (every block stays resident).
template <int NJ, bool PER = false, int PL = 2, int WV = 4>
int launch_ws_nj(const SplitMMParams& p, hipStream_t s) {
  constexpr int BN = 16 * WV * NJ;
  static int cus[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!cus[dev & 63]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return FF3D_ERR_LAUNCH;
    cus[dev & 63] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int n_tiles = (p.N + BN - 1) / BN, m_tiles = PER ? p.nbatch * ((p.period + 127) / 128) : (p.M + 127) / 128;
  // resident blocks per CU: the one-plane instance with 128-column tiles (228 registers, 64 KiB of LDS) fits twice
  // (FF3D_GEMM_WS_BF16_OCC=2, an A/B hook: the split form and the 256-column form need the whole CU)
  static const int occ1 = [] {
    const char* e = getenv("FF3D_GEMM_WS_BF16_OCC");
    return e ? atoi(e) : 1;
  }();
  const int occ = (PL == 1 && NJ == 2 && WV == 4 && occ1 == 2) ? 2 : 1;
  int groups = cus[dev & 63] * occ / n_tiles;
  if (groups < 1) groups = 1;
  if (groups > m_tiles) groups = m_tiles;
  const dim3 grid((unsigned)(groups * n_tiles)), block(64 * WV);
  constexpr size_t lds_bytes = 4 * PL * 128 * 2 * SM_BK * sizeof(_Float16) + BN * sizeof(float);  // 128 KiB (64: one plane) ring + bias tile
  ff3d_clear_error();
#define FF3D_WS(KS)                                                                                                       \
  do {                                                                                                                    \
    static bool configured[64] = {};                                                                                      \
    if (!configured[dev & 63]) {                                                                                          \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&splitmm_ws_kernel<KS, NJ, 0, PER, PL, WV>),                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)                  \
        return FF3D_ERR_LAUNCH;                                                                                           \
      configured[dev & 63] = true;                                                                                        \
    }                                                                                                                     \
    hipLaunchKernelGGL((splitmm_ws_kernel<KS, NJ, 0, PER, PL, WV>), grid, block, lds_bytes, s, p, groups);                \
  } while (0)
#ifdef FF3D_BUILD_EXPERIMENTS
  static const int abl = [] {                     // timing ablations (tuning only): FF3D_WS_ABLATE = bit mask, K = 256 only
    const char* e = getenv("FF3D_WS_ABLATE");
    return e ? atoi(e) : 0;
  }();
  if (abl && p.K == 256 && !PER && PL == 2 && WV == 4) {
#define FF3D_WSA(n)                                                                                                      \
  case n:                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&splitmm_ws_kernel<8, NJ, n>),                               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                               \
    hipLaunchKernelGGL((splitmm_ws_kernel<8, NJ, n>), grid, block, lds_bytes, s, p, groups);                             \
    break;
    switch (abl) { FF3D_WSA(1) FF3D_WSA(2) FF3D_WSA(4) FF3D_WSA(8) FF3D_WSA(6) FF3D_WSA(7) FF3D_WSA(9) FF3D_WSA(10) FF3D_WSA(12) default: break; }
#undef FF3D_WSA
    return ff3d_launch_status();
  }
#endif  // FF3D_BUILD_EXPERIMENTS
  if (p.K == 256)
    FF3D_WS(8);
  else if (p.K == 128)
    FF3D_WS(4);
  else
    return FF3D_ERR_UNSUPPORTED;
#undef FF3D_WS
  return ff3d_launch_status();
}

int launch_ws(const SplitMMParams& p, hipStream_t s) {
  if (!p.a_lo) {
    // one-plane bf16 instance.  FF3D_GEMM_WS_BF16_NJ = 2 | 4 (default 4 when the 256-column tiles divide N: weights 128 + accumulators
    // 128 registers, A streamed N / 256 times instead of N / 128)
    static const int nj1 = [] {
      const char* e = getenv("FF3D_GEMM_WS_BF16_NJ");
      return e ? atoi(e) : 4;
    }();
    if (nj1 == 4 && p.N % 256 == 0) return launch_ws_nj<4, false, 1>(p, s);
    return launch_ws_nj<2, false, 1>(p, s);
  }
  // FF3D_GEMM_WS_NJ=3: 192-column blocks when they tile N exactly (N = 768: 4 instead of 6 passes over A).  Opt-in, tuning only:
  // at K = 256 the 192 weight + 192 accumulator registers spill (75 registers) and the launch takes 5.0 ms against 2.07
  // (profiles/r03_m_ws_ab.txt); K = 128 fits.
  // The periodic (row-bias table) instance PER - 4.32 ms for both decoder stages against 2 x 2.04, slower at 1 - 4 frames
  // (profiles/r03_t_fused_value_ab.txt) - and the 192-column instance are compiled only with FF3D_BUILD_EXPERIMENTS=1; without
  // them the periodic GEMM of ff3d_gemm_f16x3_rowbias runs on the tile-streaming kernel.
#ifdef FF3D_BUILD_EXPERIMENTS
  static const int nj = [] {
    const char* e = getenv("FF3D_GEMM_WS_NJ");
    return e ? atoi(e) : 0;
  }();
  if (p.period) return launch_ws_nj<2, true>(p, s);
  if (nj == 3 && p.N % 192 == 0) return launch_ws_nj<3>(p, s);
#endif
#ifdef FF3D_BUILD_EXPERIMENTS
  // FF3D_GEMM_WS_WAVES = 8: eight waves x 16 columns (two waves per SIMD) instead of four x 32.  Round 5, measured: SLOWER, 2.87 -
  // 2.90 vs 2.05 - 2.09 ms for the value GEMM at 32 frames, 0.216 vs 0.179 ms for the neck's 1x1 layers
  // (profiles/r05_m_ws_eight_waves_ab.txt): the second wave per SIMD does not pay for 8 waves reading every A slot.
  static const int waves = [] {
    const char* e = getenv("FF3D_GEMM_WS_WAVES");
    return e ? atoi(e) : 4;
  }();
  if (waves == 8) return launch_ws_nj<1, false, 2, 8>(p, s);
#endif
#ifdef FF3D_BUILD_EXPERIMENTS
  // round 6: 256 columns per block (splitmm_ws1_kernel) for the plain fp32-output GEMM at K = 256 - measured slower, opt-in
  static const bool ws1 = [] {
    const char* e = getenv("FF3D_GEMM_WS1");
    return e && e[0] == '1';
  }();
  if (ws1 && p.K == 256 && p.N % 256 == 0 && p.out_mode == 0 && !p.res_hi && !p.period) return launch_ws1(p, s);
#endif
  return launch_ws_nj<2>(p, s);
}

int launch(const SplitMMParams& p, hipStream_t s) {
  static const int forced = [] {
    const char* e = getenv("FF3D_SPLITMM_VARIANT");     // tuning hook: "4" = the 256x128 / 3-buffer instance
    return e ? atoi(e) : 0;
  }();
  // Transposed accumulators (TR): measured on MI355X, the fp32 row-major GEMM output is SLOWER with them (value_proj 3.28
  // vs 2.89 ms at batch 32: a 16-byte store per lane still lands as 64-byte row segments, now 16 rows per instruction
  // instead of 4), so they are used for the NHWC pair outputs only (8-byte stores, no lane exchange); FF3D_TR = "all" / "none"
  // / "pair" (default) selects for experiments.
  static const int tr_mode = [] {
    const char* e = getenv("FF3D_TR");
    return !e ? 1 : (e[0] == 'a' ? 2 : e[0] == 'n' ? 0 : 1);
  }();
  // weight-stationary form: plain fp32-output GEMM with K = 128 / 256 and enough M-tiles per resident block to amortise the
  // register-resident weights - measured faster from M = 42 525 (batch 1: 75 vs 83 us) up (batch 32: 2.18 vs 2.91 ms);
  // tuning hooks: FF3D_GEMM_WS=0 disables, FF3D_GEMM_WS_MINM moves the threshold
  static const int ws_mode = [] {
    const char* e = getenv("FF3D_GEMM_WS");
    return !(e && e[0] == '0');
  }();
  static const long long ws_min_m = [] {          // tuning hook: FF3D_GEMM_WS_MINM
    const char* e = getenv("FF3D_GEMM_WS_MINM");
    return e ? atoll(e) : 32ll * 1024;
  }();
#ifdef FF3D_BUILD_EXPERIMENTS
  const bool ws_takes_period = true;              // (periodic GEMM with its row-bias table: the PER instance)
#else
  const bool ws_takes_period = false;
#endif
  const bool one_plane = p.a_lo == nullptr;        // bf16 instances (ff3d_gemm_bf16)
  static const bool ws_pair = [] {                 // pair outputs on the weight-stationary kernel too (FF3D_GEMM_WS_PAIR=0: tile-streaming)
    const char* e = getenv("FF3D_GEMM_WS_PAIR");
    return !(e && e[0] == '0');
  }();
  if (ws_mode && !p.conv && (p.out_mode == 0 || (one_plane && p.out_mode == 3) || (!one_plane && p.out_mode == 2 && ws_pair)) &&
      p.ksplit <= 1 && (!p.res_hi || (ws_pair && !one_plane)) &&
      (!p.period == !p.bias_tab) && (ws_takes_period || !p.period) && (p.K == 128 || p.K == 256) && (long long)p.M >= ws_min_m)
    return launch_ws(p, s);
  if (one_plane) {
    const long long tiles1 = (((long long)p.M + 127) / 128) * ((p.N + SM_BN - 1) / SM_BN) * (p.ksplit > 1 ? p.ksplit : 1);
    if (p.out_mode == 3) return tiles1 <= 256 ? launch_variant<2, 4, true, 1>(p, s) : launch_variant<2, 2, true, 1>(p, s);
    return tiles1 <= 256 ? launch_variant<2, 4, false, 1>(p, s) : launch_variant<2, 2, false, 1>(p, s);
  }
#ifdef FF3D_BUILD_EXPERIMENTS
  if (forced == 4) return launch_variant<4, 3, false>(p, s);
#else
  (void)forced;
#endif
  const bool tr = (p.out_mode == 2 && tr_mode >= 1) || (p.out_mode == 0 && tr_mode == 2);
  // small grids: the deep-prefetch instance (FF3D_SPLITMM_DEEP=0: never, 1: always)
  static const int deep_mode = [] {
    const char* e = getenv("FF3D_SPLITMM_DEEP");
    return e ? atoi(e) : -1;
  }();
  const long long m_tiles = p.period ? (long long)p.nbatch * ((p.period + 127) / 128) : ((long long)p.M + 127) / 128;
  const long long grid = m_tiles * ((p.N + SM_BN - 1) / SM_BN) * (p.ksplit > 1 ? p.ksplit : 1);
  if (deep_mode == 1 || (deep_mode != 0 && grid <= 256)) return tr ? launch_variant<2, 4, true>(p, s) : launch_variant<2, 4, false>(p, s);
  return tr ? launch_variant<2, 2, true>(p, s) : launch_variant<2, 2, false>(p, s);
}

}  // namespace

extern "C" int ff3d_split_f16(const float* x, void* hi, void* lo, int B, int C, int HW, int to_nhwc, int32_t* hint,
                              int32_t* out_exp, ff3d_stream_t stream) {
  FF3D_REQUIRE(x && hi && lo, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && C > 0 && HW > 0 && B <= 65535, FF3D_ERR_BAD_SHAPE);
  hipStream_t s = static_cast<hipStream_t>(stream);
  _Float16 *h = static_cast<_Float16*>(hi), *l = static_cast<_Float16*>(lo);
  ff3d_clear_error();
  const int passes = hint ? 2 : 1;                  // pass 1 = the guarded redo (exits at once when the guess held)
  if (to_nhwc) {
    const int vec4 = (HW % 4 == 0) && (C % 4 == 0) && ff3d_aligned16(x) && (reinterpret_cast<uintptr_t>(hi) % 8 == 0) &&
                     (reinterpret_cast<uintptr_t>(lo) % 8 == 0);
    const int order = split_order(C, HW, B);
    const dim3 grid = split_grid(C, HW, B, order);
    for (int pass = 0; pass < passes; ++pass) {
      hipLaunchKernelGGL(split_nchw_to_nhwc_kernel, grid, dim3(256), 0, s, x, h, l, C, HW, vec4, hint, pass, order);
      if (hint && pass == 0) hipLaunchKernelGGL(split_verify_kernel, dim3(1), dim3(64), 0, s, hint, out_exp);
    }
  } else {
    const long long n = (long long)B * C * HW;
    FF3D_REQUIRE(n % 4 == 0 && ff3d_aligned16(x), FF3D_ERR_ALIGNMENT);
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    for (int pass = 0; pass < passes; ++pass) {
      hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, h, l, n / 4, hint, pass);
      if (hint && pass == 0) hipLaunchKernelGGL(split_verify_kernel, dim3(1), dim3(64), 0, s, hint, out_exp);
    }
  }
  return ff3d_launch_status();
}

extern "C" int ff3d_split_f16_nhwc_group(int n, const float* const* x, void* const* hi, void* const* lo, int B, int C, int HW,
                                         int32_t* const* hint, int32_t* const* out_exp, ff3d_stream_t stream) {
  FF3D_REQUIRE(n >= 1 && n <= SPLIT_MAX_GROUP && x && hi && lo && hint && out_exp, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && C > 0 && HW > 0 && (long long)B * n <= 65535, FF3D_ERR_BAD_SHAPE);
  SplitGroup gp{};
  gp.B = B;
  int vec4 = (HW % 4 == 0) && (C % 4 == 0);
  for (int g = 0; g < n; ++g) {
    FF3D_REQUIRE(x[g] && hi[g] && lo[g] && hint[g] && out_exp[g], FF3D_ERR_NULL);
    gp.x[g] = x[g], gp.hi[g] = static_cast<_Float16*>(hi[g]), gp.lo[g] = static_cast<_Float16*>(lo[g]);
    gp.hint[g] = hint[g], gp.out_exp[g] = out_exp[g];
    vec4 = vec4 && ff3d_aligned16(x[g]) && (reinterpret_cast<uintptr_t>(hi[g]) % 8 == 0) && (reinterpret_cast<uintptr_t>(lo[g]) % 8 == 0);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  ff3d_clear_error();
  const int order = split_order(C, HW, (long long)B * n);
  const dim3 grid = split_grid(C, HW, B * n, order);
  for (int pass = 0; pass < 2; ++pass) {          // pass 1 = the guarded redo (each member exits at once when its guess held)
    hipLaunchKernelGGL(split_nchw_to_nhwc_group_kernel, grid, dim3(256), 0, s, gp, C, HW, vec4, pass, order);
    if (pass == 0) hipLaunchKernelGGL(split_verify_group_kernel, dim3(n), dim3(64), 0, s, gp);
  }
  return ff3d_launch_status();
}

static int conv_launch(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                       int apply_relu, float* out, void* out_hi, void* out_lo, int B, int C, int H, int W, int N,
                       int stride, const ff3d_scale_t* scale, ff3d_stream_t stream) {
  FF3D_REQUIRE(x_hi && x_lo && w_hi && w_lo && (out || (out_hi && out_lo)), FF3D_ERR_NULL);
  FF3D_REQUIRE(out || N % 2 == 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(B > 0 && C > 0 && C % SM_BK == 0 && H > 0 && W > 0 && N > 0 && (stride == 1 || stride == 2),
               FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(!scale || !scale->out_exp || scale->w_bound, FF3D_ERR_NULL);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;    // kernel 3, padding 1
  FF3D_REQUIRE((long long)B * Ho * Wo < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  // per-lane byte offsets are 32-bit: a plane (incl. its zero row) must stay below 4 GiB
  FF3D_REQUIRE(((long long)B * H * W + 1) * C * 2 < (1ll << 32) && ((long long)N + 1) * 9 * C * 2 < (1ll << 32),
               FF3D_ERR_BAD_SHAPE);
  // round 6: the stride-2 convs of the BEV pyramid (N = 256 output channels, fp32 NCHW result) with SWAPPED operands - the weight is the
  // row operand (one 256 x 128 block covers every output channel), the activation's implicit-GEMM gather sits on the 128-wide operand:
  // each activation tile is gathered once instead of once per 128-channel tile (PMC, round 5: 2 x 1.32 GB fetched for a 1.06 GB input).
  // FF3D_CONV_S2_SWAP=0: the 128 x 128 tiles of rounds 1-5 (A/B: profiles/r06_q_conv_s2_swap_ab.txt).
  static const bool s2_swap = [] {
    const char* e = getenv("FF3D_CONV_S2_SWAP");
    return !(e && e[0] == '0');
  }();
  if (s2_swap && stride == 2 && out && N == 256 && (long long)B * Ho * Wo >= 128) {
    SplitMMParams q{static_cast<const _Float16*>(w_hi), static_cast<const _Float16*>(w_lo),
                    static_cast<const _Float16*>(x_hi), static_cast<const _Float16*>(x_lo), bias, out, nullptr, nullptr, nullptr,
                    nullptr, INFINITY, N, B * Ho * Wo, 9 * C, 0, C, H, W, Ho, Wo, stride, apply_relu ? 1 : 0, 1, 1,
                    (unsigned)((long long)N * 9 * C * 2), (unsigned)((long long)B * H * W * C * 2), ff3d_scale_from(scale), 0, 0, nullptr};
    q.conv_b = 1;
    return launch_variant<4, 3, true, 2, true>(q, static_cast<hipStream_t>(stream));
  }
  SplitMMParams p{static_cast<const _Float16*>(x_hi), static_cast<const _Float16*>(x_lo),
                  static_cast<const _Float16*>(w_hi), static_cast<const _Float16*>(w_lo), bias, out,
                  static_cast<_Float16*>(out_hi), static_cast<_Float16*>(out_lo), nullptr, nullptr, INFINITY,
                  B * Ho * Wo, N, 9 * C, 1, C, H, W, Ho, Wo, stride, apply_relu ? 1 : 0, out ? 1 : 2, 1,
                  (unsigned)((long long)B * H * W * C * 2), (unsigned)((long long)N * 9 * C * 2), ff3d_scale_from(scale), 0, 0, nullptr};
  return launch(p, static_cast<hipStream_t>(stream));
}

extern "C" int ff3d_conv3x3_f16x3(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo,
                                  const float* bias, int apply_relu, float* out, int B, int C, int H, int W, int N,
                                  int stride, const ff3d_scale_t* scale_host, ff3d_stream_t stream) {
  FF3D_REQUIRE(out, FF3D_ERR_NULL);
  return conv_launch(x_hi, x_lo, w_hi, w_lo, bias, apply_relu, out, nullptr, nullptr, B, C, H, W, N, stride, scale_host,
                     stream);
}

extern "C" int ff3d_conv3x3_f16x3_split_out(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo,
                                            const float* bias, int apply_relu, void* out_hi, void* out_lo, int B,
                                            int C, int H, int W, int N, int stride, const ff3d_scale_t* scale_host,
                                            ff3d_stream_t stream) {
  return conv_launch(x_hi, x_lo, w_hi, w_lo, bias, apply_relu, nullptr, out_hi, out_lo, B, C, H, W, N, stride, scale_host,
                     stream);
}

// Round 6 (third session): the fp32-output conv with its K = 9 * C walk cut into `ksplit` slices (the BEV pyramid's stride-2 convs at one
// to four frames, FD:150-162: 90 x 90 / 45 x 45 output pixels are 64 / 16 row tiles - a handful of blocks that each walk 72 K-steps,
// 107 us where the arithmetic needs 7 - 29).  Slice s of a row tile is one more block (grid.y) that starts inside the tap walk and
// writes raw partial sums to plane s of `workspace` (ksplit, B*Ho*Wo, N); splitk_reduce_nchw_kernel adds the planes in slice order,
// applies scale / bias / ReLU and writes the NCHW map.  Deterministic for a given ksplit; the K order of the fp32 sum differs from the
// one-pass kernel's (both are fp32-class: tests/test_round6_gpu.py against fp64).
extern "C" int ff3d_conv3x3_f16x3_splitk(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                                         int apply_relu, float* out, int B, int C, int H, int W, int N, int stride, int ksplit,
                                         float* workspace, const ff3d_scale_t* scale, ff3d_stream_t stream) {
  FF3D_REQUIRE(x_hi && x_lo && w_hi && w_lo && out && workspace, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && C > 0 && C % SM_BK == 0 && H > 0 && W > 0 && N > 0 && (stride == 1 || stride == 2), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ksplit >= 2 && ksplit <= 64 && ksplit <= 9 * C / SM_BK, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(workspace) && ff3d_aligned16(out), FF3D_ERR_ALIGNMENT);
  FF3D_REQUIRE(!scale || !scale->out_exp || scale->w_bound, FF3D_ERR_NULL);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  FF3D_REQUIRE((long long)B * Ho * Wo < (1ll << 31) && (long long)B * Ho * Wo * N * ksplit < (1ll << 40), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(((long long)B * H * W + 1) * C * 2 < (1ll << 32) && ((long long)N + 1) * 9 * C * 2 < (1ll << 32), FF3D_ERR_BAD_SHAPE);
  const int M = B * Ho * Wo;
  Ff3dScale sc = ff3d_scale_from(scale), sc_main = sc;
  sc_main.out_exp = nullptr;                        // written by the reduce kernel
  SplitMMParams p{static_cast<const _Float16*>(x_hi), static_cast<const _Float16*>(x_lo),
                  static_cast<const _Float16*>(w_hi), static_cast<const _Float16*>(w_lo), nullptr, workspace, nullptr, nullptr,
                  nullptr, nullptr, INFINITY, M, N, 9 * C, 1, C, H, W, Ho, Wo, stride, 0, 0, ksplit,
                  (unsigned)((long long)B * H * W * C * 2), (unsigned)((long long)N * 9 * C * 2), sc_main, 0, 0, nullptr};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int st = launch(p, s);
  if (st != FF3D_OK) return st;
  hipLaunchKernelGGL(splitk_reduce_nchw_kernel, dim3((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64)), dim3(256), 0, s, workspace,
                     bias, out, M, N, Ho * Wo, ksplit, apply_relu ? 1 : 0, INFINITY, sc);
  return ff3d_launch_status();
}

extern "C" int ff3d_gemm_f16x3_fused(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo,
                                     const float* bias, int act, const void* res_hi, const void* res_lo, float* out,
                                     void* out_hi, void* out_lo, int M, int N, int K, int ksplit, float* workspace,
                                     const ff3d_scale_t* scale_host, ff3d_stream_t stream) {
  FF3D_REQUIRE(a_hi && a_lo && w_hi && w_lo && (out || (out_hi && out_lo)), FF3D_ERR_NULL);
  FF3D_REQUIRE(M > 0 && N > 0 && K > 0 && K % SM_BK == 0 && act >= 0 && act <= 2, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(out || N % 2 == 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(!res_hi == !res_lo, FF3D_ERR_NULL);
  FF3D_REQUIRE(ksplit >= 1 && ksplit <= 64 && ksplit <= K / SM_BK, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ksplit == 1 || (workspace && !res_hi && out && ff3d_aligned16(workspace) && ff3d_aligned16(out)),
               FF3D_ERR_NULL);
  FF3D_REQUIRE(!scale_host || !scale_host->out_exp || scale_host->w_bound, FF3D_ERR_NULL);
  FF3D_REQUIRE(((long long)M + 1) * K * 2 < (1ll << 32) && ((long long)N + 1) * K * 2 < (1ll << 32), FF3D_ERR_BAD_SHAPE);
  const float upper = act == 2 ? 6.f : INFINITY;
  Ff3dScale sc = ff3d_scale_from(scale_host), sc_main = sc;
  if (ksplit > 1) sc_main.out_exp = nullptr;       // written by the reduce kernel
  SplitMMParams p{static_cast<const _Float16*>(a_hi), static_cast<const _Float16*>(a_lo),
                  static_cast<const _Float16*>(w_hi), static_cast<const _Float16*>(w_lo), bias, ksplit > 1 ? workspace : out,
                  static_cast<_Float16*>(out_hi), static_cast<_Float16*>(out_lo),
                  static_cast<const _Float16*>(res_hi), static_cast<const _Float16*>(res_lo), upper,
                  M, N, K, 0, 0, 0, 0, 1, M, 1, act ? 1 : 0, out ? 0 : 2, ksplit, (unsigned)((long long)M * K * 2),
                  (unsigned)((long long)N * K * 2), sc_main, 0, 0, nullptr};
  hipStream_t s = static_cast<hipStream_t>(stream);
  // long-K GEMM with few output columns and many rows (roi_mlp.0: 19 200 x 37 632 x 512): the 128-column tiles make N / 128 blocks
  // stream every activation tile (PMC FETCH_SIZE 7.6 GB per launch for a 2.9 GB panel, profiles/r05_l_pmc_l_*); with the operands
  // swapped on the 256 x 128 instance it is N / 256.  FF3D_GEMM_SWAP=0: never.
  static const bool swap_ok = [] {
    const char* e = getenv("FF3D_GEMM_SWAP");
    return !(e && e[0] == '0');
  }();
  // (round 6, third session: from 512 rows - one frame - instead of 4 096: one-frame replay 1.855 vs 1.870 ms on the device, the
  // pipelined 4-frame step 3.274 vs 3.302 ms, two alternations each: profiles/r06_sw5_roi_mlp_swap_small_m.txt)
  static const int swap_min_m = [] {               // tuning hook: FF3D_GEMM_SWAP_MINM
    const char* e = getenv("FF3D_GEMM_SWAP_MINM");
    return e ? atoi(e) : 512;
  }();
  int st;
  if (swap_ok && ksplit > 1 && N % 256 == 0 && N <= 1024 && M >= swap_min_m && M % 4 == 0) {
    SplitMMParams q = p;
    q.a_hi = p.w_hi, q.a_lo = p.w_lo, q.w_hi = p.a_hi, q.w_lo = p.a_lo;
    q.M = N, q.N = M, q.Wo = N;                       // (Ho x Wo = the GEMM's row count in the non-conv form)
    q.a_zero = p.b_zero, q.b_zero = p.a_zero;
    q.sc.a_exp = p.sc.w_exp, q.sc.w_exp = p.sc.a_exp;
    q.bias = nullptr, q.swap_out = 1;
    st = launch_variant<4, 3, false>(q, s);
  } else {
    st = launch(p, s);
  }
  if (st != FF3D_OK || ksplit == 1) return st;
  const long long MN = (long long)M * N;
  long long blocks = (MN / 4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, workspace, bias, out, MN, N, ksplit,
                     act ? 1 : 0, upper, sc);
  return ff3d_launch_status();
}

extern "C" int ff3d_gemm_f16x3(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo,
                               const float* bias, int apply_relu, float* out, int M, int N, int K, int ksplit,
                               float* workspace, const ff3d_scale_t* scale_host, ff3d_stream_t stream) {
  FF3D_REQUIRE(out, FF3D_ERR_NULL);
  return ff3d_gemm_f16x3_fused(a_hi, a_lo, w_hi, w_lo, bias, apply_relu ? 1 : 0, nullptr, nullptr, out, nullptr, nullptr,
                               M, N, K, ksplit, workspace, scale_host, stream);
}

// value_proj of every decoder stage / layer in one launch: out[b*rows + r, n] = sum_k A[b*rows + r, k] W[n, k] + tab[r, n].
// The table carries everything that does not depend on the frame: tab = pos_embed @ W^T + bias (mmcv MSDA computes
// value_proj(feats + bev_pos_embed), FD:886 + the first line of MultiScaleDeformableAttention.forward; by linearity the
// positional term moves into this weight-only table).
extern "C" int ff3d_gemm_f16x3_rowbias(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo,
                                       const float* bias_tab, float* out, int nbatch, int rows, int N, int K,
                                       const ff3d_scale_t* scale_host, ff3d_stream_t stream) {
  FF3D_REQUIRE(a_hi && a_lo && w_hi && w_lo && bias_tab && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(nbatch > 0 && rows > 0 && N > 0 && K > 0 && K % SM_BK == 0, FF3D_ERR_BAD_SHAPE);
  const long long M = (long long)nbatch * rows;
  FF3D_REQUIRE(M < (1ll << 31) && (M + 1) * K * 2 < (1ll << 32) && ((long long)N + 1) * K * 2 < (1ll << 32), FF3D_ERR_BAD_SHAPE);
  SplitMMParams p{static_cast<const _Float16*>(a_hi), static_cast<const _Float16*>(a_lo),
                  static_cast<const _Float16*>(w_hi), static_cast<const _Float16*>(w_lo), nullptr, out, nullptr, nullptr,
                  nullptr, nullptr, INFINITY, (int)M, N, K, 0, 0, 0, 0, 1, (int)M, 1, 0, 0, 1, (unsigned)(M * K * 2),
                  (unsigned)((long long)N * K * 2), ff3d_scale_from(scale_host), rows, nbatch, bias_tab};
  p.sc.out_exp = nullptr;
  return launch(p, static_cast<hipStream_t>(stream));
}

// bf16 GEMM of BASELINE configs[4] ("bf16 QKV/FFN on MFMA"): out (M, N) = act(A (M, K) bf16 @ W (N, K)^T bf16 + bias) with exact
// products, fp32 accumulation, bias in fp32, ONE rounding to bf16, ReLU on the rounded value (oracle/ff3d_oracle.py lin(lowp=True)).
// Both operand planes end with one zero row (ZERO-ROW CONTRACT).  Result: `out` (fp32 rows holding bf16 values) or `out_bf16` (bf16
// rows) - exactly one.  K = 128 / 256 and M >= 32 768: the weight-stationary kernel (value_proj, FD:886 + mmcv MSDA.forward);
// otherwise the tile-streaming kernel, with `ksplit` K-slices through `workspace` (ksplit, M, N) for long K (roi_mlp.0, FD:186-200).
extern "C" int ff3d_gemm_bf16(const void* a, const void* w, const float* bias, int apply_relu, float* out, void* out_bf16, int M, int N,
                              int K, int ksplit, float* workspace, ff3d_stream_t stream) {
  FF3D_REQUIRE(a && w && (!out != !out_bf16), FF3D_ERR_NULL);
  FF3D_REQUIRE(M > 0 && N > 0 && K > 0 && K % SM_BK == 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ksplit >= 1 && ksplit <= 64 && ksplit <= K / SM_BK, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ksplit == 1 || (workspace && out && ff3d_aligned16(workspace) && ff3d_aligned16(out)), FF3D_ERR_NULL);
  FF3D_REQUIRE(((long long)M + 1) * K * 2 < (1ll << 32) && ((long long)N + 1) * K * 2 < (1ll << 32), FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(a) && ff3d_aligned16(w) && (!out_bf16 || N % 4 == 0), FF3D_ERR_ALIGNMENT);
  Ff3dScale sc{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  SplitMMParams p{static_cast<const _Float16*>(a), nullptr, static_cast<const _Float16*>(w), nullptr, bias,
                  ksplit > 1 ? workspace : out, static_cast<_Float16*>(out_bf16), nullptr, nullptr, nullptr, INFINITY,
                  M, N, K, 0, 0, 0, 0, 1, M, 1, apply_relu ? 1 : 0, out ? 0 : 3, ksplit, (unsigned)((long long)M * K * 2),
                  (unsigned)((long long)N * K * 2), sc, 0, 0, nullptr};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int st = launch(p, s);
  if (st != FF3D_OK || ksplit == 1) return st;
  const long long MN = (long long)M * N;
  long long blocks = (MN / 4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, workspace, bias, out, MN, N, ksplit,
                     apply_relu ? 1 : 0, INFINITY, sc, 1);
  return ff3d_launch_status();
}
