// LSS pillar pooling (interval sums) for gfx950 - counterpart of the reference's CUDA extension
// projects/mmdet3d_plugin/models/utils/ops/bev_pool (src/bev_pool_cuda.cu:20-42, launch :86-91).
//
// Points arrive sorted by BEV cell rank; every interval [start, start+len) of the (n, c) feature matrix is summed
// into one (b, z, x, y) cell of the channels-last output.  The reference gives one THREAD per (interval, channel)
// and walks the interval serially; here one 64-lane wave owns an interval, lanes cover 4 channels each with 16-byte
// loads (c = 80 -> 20 active lanes x float4, rows fully coalesced), rows are unrolled by 4 so several row loads are
// in flight, and a wave grid-strides over intervals.  HBM-bound: every input element is read exactly once
// (algorithmic bytes = n*c*4 + n_intervals*(c*4 + 24)).
#include "ff3d_common.h"

namespace {

__global__ __launch_bounds__(256) void bev_pool_kernel(int d, int h, int w, int c, int n_intervals,
                                                       const float* __restrict__ x, const int* __restrict__ geom,
                                                       const int* __restrict__ starts, const int* __restrict__ lengths,
                                                       float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
  const int c4 = c >> 2;
  for (int it = wave; it < n_intervals; it += nwaves) {
    const int s = starts[it], len = lengths[it];
    const int* g = geom + (long long)s * 4;              // (x, y, z, b) of the interval's first point
    float* o = out + ((((long long)g[3] * d + g[2]) * h + g[0]) * w + g[1]) * c;
    for (int cc = lane; cc < c4; cc += 64) {
      const float4* px = reinterpret_cast<const float4*>(x + (long long)s * c) + cc;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int i = 0;
      for (; i + 4 <= len; i += 4) {                     // same summation order as the reference (row by row)
        const float4 a = px[(long long)i * c4], b4 = px[(long long)(i + 1) * c4];
        const float4 e = px[(long long)(i + 2) * c4], f = px[(long long)(i + 3) * c4];
        acc.x = ((acc.x + a.x) + b4.x) + e.x + f.x;
        acc.y = ((acc.y + a.y) + b4.y) + e.y + f.y;
        acc.z = ((acc.z + a.z) + b4.z) + e.z + f.z;
        acc.w = ((acc.w + a.w) + b4.w) + e.w + f.w;
      }
      for (; i < len; ++i) {
        const float4 a = px[(long long)i * c4];
        acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
      }
      reinterpret_cast<float4*>(o)[cc] = acc;
    }
  }
}

// Backward (bev_pool_cuda.cu:61-84 bev_pool_grad_kernel): every point of an interval receives the gradient row of the
// interval's cell.  One wave per interval again: the (c) gradient row is read once (16-byte lanes) and written to the
// interval's `len` rows of x_grad - each output element written exactly once, nothing to zero, deterministic.
__global__ __launch_bounds__(256) void bev_pool_bwd_kernel(int d, int h, int w, int c, int n_intervals,
                                                           const float* __restrict__ out_grad, const int* __restrict__ geom,
                                                           const int* __restrict__ starts, const int* __restrict__ lengths,
                                                           float* __restrict__ x_grad) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
  const int c4 = c >> 2;
  for (int it = wave; it < n_intervals; it += nwaves) {
    const int s = starts[it], len = lengths[it];
    const int* g = geom + (long long)s * 4;
    const float* o = out_grad + ((((long long)g[3] * d + g[2]) * h + g[0]) * w + g[1]) * c;
    for (int cc = lane; cc < c4; cc += 64) {
      const float4 v = reinterpret_cast<const float4*>(o)[cc];
      float4* px = reinterpret_cast<float4*>(x_grad + (long long)s * c) + cc;
      for (int i = 0; i < len; ++i) px[(long long)i * c4] = v;
    }
  }
}

}  // namespace

extern "C" int ff3d_bev_pool_bwd(const float* out_grad, const int32_t* geom_feats, const int32_t* interval_starts,
                                 const int32_t* interval_lengths, float* x_grad, int b, int d, int h, int w, int n, int c,
                                 int n_intervals, ff3d_stream_t stream) {
  FF3D_REQUIRE(out_grad && geom_feats && interval_starts && interval_lengths && x_grad, FF3D_ERR_NULL);
  FF3D_REQUIRE(b > 0 && d > 0 && h > 0 && w > 0 && n > 0 && c > 0 && n_intervals > 0 && n_intervals <= n,
               FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(c % 4 == 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(out_grad) && ff3d_aligned16(x_grad), FF3D_ERR_ALIGNMENT);
  int blocks = (n_intervals + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  ff3d_clear_error();
  hipLaunchKernelGGL(bev_pool_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d, h, w, c,
                     n_intervals, out_grad, geom_feats, interval_starts, interval_lengths, x_grad);
  return ff3d_launch_status();
}

extern "C" int ff3d_bev_pool(const float* x, const int32_t* geom_feats, const int32_t* interval_starts,
                             const int32_t* interval_lengths, float* out, int b, int d, int h, int w, int n, int c,
                             int n_intervals, ff3d_stream_t stream) {
  FF3D_REQUIRE(x && geom_feats && interval_starts && interval_lengths && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(b > 0 && d > 0 && h > 0 && w > 0 && n > 0 && c > 0 && n_intervals > 0 && n_intervals <= n,
               FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(c % 4 == 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(x) && ff3d_aligned16(out), FF3D_ERR_ALIGNMENT);
  const int waves_needed = n_intervals;
  int blocks = (waves_needed + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;              // grid-stride beyond 16 blocks per CU
  ff3d_clear_error();
  hipLaunchKernelGGL(bev_pool_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d, h, w, c,
                     n_intervals, x, geom_feats, interval_starts, interval_lengths, out);
  return ff3d_launch_status();
}
