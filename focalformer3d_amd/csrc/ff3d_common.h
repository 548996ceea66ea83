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
