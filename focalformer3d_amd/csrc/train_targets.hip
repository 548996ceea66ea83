// Dense heatmap targets of the head's training loss on the device: FocalDecoder.get_targets_single, FD:1133-1158 - for every
// ground-truth box the CenterNet Gaussian radius (mmdet3d `gaussian_radius`, fp32 tensor arithmetic in the reference), the
// truncated centre cell and `draw_heatmap_gaussian` (element-wise max with a (2r+1)^2 Gaussian, sigma = (2r+1)/6, values
// computed in double and rounded to fp32 as numpy -> torch does) into the class plane of the box.  The reference loops over
// the boxes on the host with one device round trip per box; here one block per box, atomicMax on the fp32 bit pattern
// (all values are >= 0, so the unsigned order is the float order and the result does not depend on the block order).
#include "ff3d_common.h"

namespace {

struct HeatTargetParams {
  const float* gt;          // (m, box_dim): x, y, z_bottom, dx, dy, dz, yaw, ...
  const long long* labels;  // (m)
  float* heatmap;           // (K, H, W), zero-initialised by the caller
  int m, box_dim, K, H, W, min_radius;
  float vx, vy, pcx, pcy, osf;
  float c_1m, c_1p, c_4a3, c_b3, c_c3;   // (1 - o), (1 + o), 4 * (4 * o), -2 * o, (o - 1) as the reference's Python floats, in fp32
};

#pragma clang fp contract(off)   // the reference evaluates these expressions op by op in fp32 (no fused multiply-add)
__device__ int gaussian_radius_fp32(float height, float width, const HeatTargetParams& p) {
  const float b1 = height + width;
  const float c1 = width * height * p.c_1m / p.c_1p;
  const float r1 = (b1 + sqrtf(b1 * b1 - 4.f * c1)) / 2.f;
  const float b2 = 2.f * (height + width);
  const float c2 = p.c_1m * width * height;
  const float r2 = (b2 + sqrtf(b2 * b2 - 16.f * c2)) / 2.f;
  const float b3 = p.c_b3 * (height + width);
  const float c3 = p.c_c3 * width * height;
  const float r3 = (b3 + sqrtf(b3 * b3 - p.c_4a3 * c3)) / 2.f;
  return (int)fminf(r1, fminf(r2, r3));
}

__global__ __launch_bounds__(256) void gaussian_heatmap_targets_kernel(HeatTargetParams p) {
  const int i = blockIdx.x;
  const float* g = p.gt + (long long)i * p.box_dim;
  const float width = g[3] / p.vx / p.osf, length = g[4] / p.vy / p.osf;
  if (!(width > 0.f && length > 0.f)) return;
  const int radius = max(p.min_radius, gaussian_radius_fp32(length, width, p));
  const int x = (int)((g[0] - p.pcx) / p.vx / p.osf), y = (int)((g[1] - p.pcy) / p.vy / p.osf);   // .to(int32): truncation
  const long long cls = p.labels[i];
  if (cls < 0 || cls >= p.K) return;
  float* plane = p.heatmap + cls * (long long)p.H * p.W;
  const int d = 2 * radius + 1;
  const double sigma = (double)d / 6.0, inv = 1.0 / (2.0 * sigma * sigma);
  for (int e = threadIdx.x; e < d * d; e += 256) {
    const int dy = e / d - radius, dx = e % d - radius;
    const int yy = y + dy, xx = x + dx;
    if (yy < 0 || yy >= p.H || xx < 0 || xx >= p.W) continue;
    const float v = (float)exp(-(double)(dx * dx + dy * dy) * inv);
    atomicMax(reinterpret_cast<unsigned*>(plane + (long long)yy * p.W + xx), __float_as_uint(v));
  }
}

}  // namespace

extern "C" int ff3d_gaussian_heatmap_targets(const float* gt_boxes, const int64_t* gt_labels, float* heatmap, int m,
                                             int box_dim, int K, int H, int W, const float* coder_host,
                                             float gaussian_overlap, int min_radius, ff3d_stream_t stream) {
  FF3D_REQUIRE(heatmap && coder_host && (m == 0 || (gt_boxes && gt_labels)), FF3D_ERR_NULL);
  FF3D_REQUIRE(m >= 0 && box_dim >= 7 && K > 0 && H > 0 && W > 0 && min_radius >= 0, FF3D_ERR_BAD_SHAPE);
  if (m == 0) return FF3D_OK;
  const double o = (double)gaussian_overlap;
  HeatTargetParams p{gt_boxes, reinterpret_cast<const long long*>(gt_labels), heatmap, m, box_dim, K, H, W, min_radius,
                     coder_host[1], coder_host[2], coder_host[3], coder_host[4], coder_host[0],
                     (float)(1.0 - o), (float)(1.0 + o), (float)(4.0 * (4.0 * o)), (float)(-2.0 * o), (float)(o - 1.0)};
  ff3d_clear_error();
  hipLaunchKernelGGL(gaussian_heatmap_targets_kernel, dim3(m), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return ff3d_launch_status();
}
