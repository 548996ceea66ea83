// Shared helpers for the gfx950 kernels of libff3d_hip.so.  wave = 64 lanes throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ff3d.h"

#define FF3D_WAVE 64
#define FF3D_NUM_XCD 8

#define FF3D_REQUIRE(cond, code) \
  do {                           \
    if (!(cond)) return (code);  \
  } while (0)

static inline bool ff3d_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// hipGetLastError() returns (and clears) the last error of ANY earlier runtime call on this thread - e.g. a
// benign query failure inside MIOpen/rocBLAS - so entry points clear it before enqueueing and read it after.
static inline void ff3d_clear_error() { (void)hipGetLastError(); }
static inline int ff3d_launch_status() { return hipGetLastError() == hipSuccess ? FF3D_OK : FF3D_ERR_LAUNCH; }

struct LevelTable {
  int H[FF3D_MAX_LEVELS];
  int W[FF3D_MAX_LEVELS];
  int start[FF3D_MAX_LEVELS];
  int L;
  int Nv;
};

// Returns false if the table is malformed.
static inline bool ff3d_make_levels(const int32_t* hw_host, int L, LevelTable* t) {
  if (!hw_host || L < 1 || L > FF3D_MAX_LEVELS) return false;
  int acc = 0;
  for (int l = 0; l < L; ++l) {
    t->H[l] = hw_host[2 * l];
    t->W[l] = hw_host[2 * l + 1];
    if (t->H[l] <= 0 || t->W[l] <= 0) return false;
    t->start[l] = acc;
    acc += t->H[l] * t->W[l];
  }
  for (int l = L; l < FF3D_MAX_LEVELS; ++l) t->H[l] = t->W[l] = t->start[l] = 0;
  t->L = L;
  t->Nv = acc;
  return true;
}

// XCD-aware block remap (blocks are dispatched round-robin over the 8 XCDs, each with a private
// 4 MiB L2): give every XCD one contiguous chunk of the logical grid so neighbouring work items
// (same frame / neighbouring queries) share an L2.  Bijective for any grid size.
__device__ __forceinline__ unsigned ff3d_xcd_remap(unsigned bid, unsigned nblocks) {
  const unsigned q = nblocks / FF3D_NUM_XCD, r = nblocks % FF3D_NUM_XCD;
  const unsigned xcd = bid % FF3D_NUM_XCD, slot = bid / FF3D_NUM_XCD;
  // XCDs [0, r) own q+1 blocks, the rest q blocks
  const unsigned base = xcd * q + (xcd < r ? xcd : r);
  return base + slot;
}

__device__ __forceinline__ float ff3d_bf16_to_f32(unsigned short v) { return __uint_as_float(((unsigned)v) << 16); }

// ---- range normalisation of split-fp16 operands (ff3d.h: RANGE NORMALISATION, ff3d_scale_t) -------------------------------
// x = 2^e * (hi + lo'/2048) with |x| * 2^-e < 2^15; exponents are int32 scalars in device memory.
struct Ff3dScale {
  const int* a_exp;
  const int* a2_exp;
  const int* w_exp;
  const float* w_bound;   // {L1(W), max|bias|}
  const int* res_exp;
  int* out_exp;
};

static inline Ff3dScale ff3d_scale_from(const ff3d_scale_t* s) {
  Ff3dScale r{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (s) {
    r.a_exp = s->a_exp, r.a2_exp = s->a2_exp, r.w_exp = s->w_exp, r.w_bound = s->w_bound, r.res_exp = s->res_exp;
    r.out_exp = s->out_exp;
  }
  return r;
}

constexpr int FF3D_EXP_TOP = 15;      // |x * 2^-e| < 2^FF3D_EXP_TOP for every scaled operand
constexpr int FF3D_EXP_TARGET = 14;   // exponents are chosen so that the bound / measured maximum lands below 2^14

__device__ __forceinline__ int ff3d_ld_exp(const int* p) { return p ? *p : 0; }
// 2^e as a float (e clamped to the normal range; exponents outside it only occur for all-zero / non-finite tensors)
__device__ __forceinline__ float ff3d_pow2(int e) { return __int_as_float((min(max(e, -126), 127) + 127) << 23); }
// smallest e with bound * 2^-e < 2^FF3D_EXP_TARGET (bound >= 0; zero / denormal bounds give a harmless small exponent)
__device__ __forceinline__ int ff3d_bound_exp(float bound) {
  const int ex = (int)((__float_as_uint(bound) >> 23) & 0xffu) - 127;   // floor(log2(bound)) for normal floats
  return ex + 1 - FF3D_EXP_TARGET;
}
// exponent of a layer output from the guaranteed bound |out| <= 2^(e_in + TOP) * L1(W) + max|bias| (+ 2^(e_res + TOP)),
// clamped by the activation's upper limit; e_in already includes every input exponent the caller wants (max over inputs).
__device__ __forceinline__ int ff3d_out_exp(const Ff3dScale& s, int e_in, bool has_res, float upper) {
  float bound = s.w_bound ? fmaf(ff3d_pow2(e_in + FF3D_EXP_TOP), s.w_bound[0], s.w_bound[1]) : ff3d_pow2(e_in + FF3D_EXP_TOP);
  if (has_res) bound += ff3d_pow2(ff3d_ld_exp(s.res_exp) + FF3D_EXP_TOP);
  bound = fminf(bound, upper);
  return ff3d_bound_exp(bound);
}
