// Micro-benchmark (round 5): how fast do all CUs stream the SAME weight slab L2 -> LDS, as a function of the slab's row stride?
// A block issues what linrows.hip's dma_w issues per K-step: 256 rows x 64 B per plane, two planes, 16-byte LDS-DMA pieces, and walks
// `steps` K-steps (64 B further along every row, or - layout "tiled" - the next contiguous 16 KiB tile).  Every block reads the same
// addresses (weights are shared by all row tiles).  Prints GB/s per layout; build: hipcc --offload-arch=gfx950 -O3 -o /tmp/slab tools/bench_l2_slab.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                            \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

__global__ __launch_bounds__(512) void slab_kernel(const char* w_hi, const char* w_lo, long long row_stride, long long step_stride,
                                                  int steps, int depth, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];      // [3 stages][2 planes][16 KiB]
  const int tid = threadIdx.x, wave = tid >> 6;
  unsigned off[2];
  for (int j = 0; j < 2; ++j) {
    const int s = j * 512 + tid, row = s >> 2;
    off[j] = (unsigned)(row * row_stride + (s & 3) * 16);
  }
  auto dma = [&](int g) {
    char* base = lds + (g % 3) * 32768;
    for (int j = 0; j < 2; ++j) {
      char* dst = base + (j * 512 + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds(w_hi + off[j] + (unsigned)(g * step_stride), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds(w_lo + off[j] + (unsigned)(g * step_stride), (__attribute__((address_space(3))) void*)(dst + 16384), 16, 0,
                                       0);
    }
  };
  float acc = 0.f;
  for (int q = 0; q < depth && q < steps; ++q) dma(q);
  for (int g = 0; g < steps; ++g) {
    if (depth == 2) {
      if (g + 1 < steps)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (g + depth < steps) dma(g + depth);
    acc += *reinterpret_cast<float*>(lds + (g % 3) * 32768 + tid * 16);      // touch the stage
  }
  if (acc == 123.456f) sink[0] = acc;
}

int main() {
  const int N = 256, steps = 32, blocks = 240, reps = 50;
  struct Case { const char* name; long long row_stride, step_stride; };
  std::vector<Case> cases = {
      {"tiled (rows 64 B apart, K-step = next 16 KiB tile)", 64, 16384},
      {"row-major K=256 (stride 512 B)", 512, 64},
      {"row-major K=1024 (stride 2 KiB)", 2048, 64},
      {"row-major K=2304 (stride 4608 B: 3x3 conv, C=256)", 4608, 64},
      {"row-major K=1056 (stride 2112 B: padded rows)", 2112, 64},
  };
  size_t bytes = (size_t)N * 4608 + (size_t)steps * 16384 + 65536;
  char *hi, *lo;
  float* sink;
  CK(hipMalloc(&hi, bytes));
  CK(hipMalloc(&lo, bytes));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(hi, 0, bytes));
  CK(hipMemset(lo, 0, bytes));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&slab_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int depth = 1; depth <= 2; ++depth)
    for (auto& c : cases) {
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(slab_kernel, dim3(blocks), dim3(512), 98304, 0, hi, lo, c.row_stride, c.step_stride, steps, depth, sink);
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(slab_kernel, dim3(blocks), dim3(512), 98304, 0, hi, lo, c.row_stride, c.step_stride, steps, depth, sink);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps, gb = (double)blocks * steps * 32768 / 1e9;
      printf("depth %d  %-56s %8.2f us/launch  %7.2f us/K-step  %8.1f GB/s L2->LDS\n", depth, c.name, us, us / steps, gb / (us * 1e-6));
    }
  return 0;
}
