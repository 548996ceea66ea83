// Rotated BEV IoU matrix and single-set rotated NMS for gfx950 - the two mmdet3d 0.17.1 iou3d ops the reference's
// test-time-augmentation merging calls (projects/mmdet3d_plugin/core/post_processing/merge_augs.py:137, 150:
// `nms_gpu`, `boxes_iou_bev`; un-vendored third party, algorithm restated in rotiou.h).
//   boxes_iou_bev_kernel: one thread per (a, b) pair, b fastest.
//   nms_bev_kernel: one block; LDS bitonic sort of (score, index) keys, sequential sweep in score order with the
//   suppression of every kept box parallel over the block (boxes stay in global memory / L2), kept indices emitted in
//   score order as `order[keep]` of nms_gpu.
#include "ff3d_common.h"
#include "rotiou.h"

namespace {

constexpr int NB_THREADS = 256, NB_MAX = 4096;

__global__ __launch_bounds__(256) void boxes_iou_bev_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float* __restrict__ out, int N, int M) {
  const long long total = (long long)N * M;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int i = (int)(e / M), j = (int)(e - (long long)i * M);
    float ba[5], bb[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) ba[k] = a[i * 5 + k], bb[k] = b[j * 5 + k];
    out[e] = ff3d_rot::iou_bev(ba, bb);
  }
}

__global__ __launch_bounds__(NB_THREADS) void nms_bev_kernel(const float* __restrict__ boxes,
                                                             const float* __restrict__ scores, float thresh,
                                                             int pre_max, int post_max, int* __restrict__ keep,
                                                             int* __restrict__ count, int n) {
  __shared__ unsigned long long keys[NB_MAX];
  __shared__ unsigned char state[NB_MAX];     // by sorted position: 0 undecided, 1 kept, 2 suppressed
  const int tid = threadIdx.x;
  int n2 = 2;
  while (n2 < n) n2 <<= 1;
  for (int i = tid; i < n2; i += NB_THREADS) {
    // descending score, ties by lower index; negative scores order correctly through the sign-flip key
    unsigned u = i < n ? __float_as_uint(scores[i]) : 0u;
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    keys[i] = i < n ? (((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (unsigned)i)) : 0ull;
    state[i] = 0;
  }
  for (int size = 2; size <= n2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = tid; i < (n2 >> 1); i += NB_THREADS) {
        const int pos = 2 * i - (i & (stride - 1)), j = pos + stride;
        const bool up = (pos & size) == 0;
        const unsigned long long x = keys[pos], y = keys[j];
        if ((x < y) == up) keys[pos] = y, keys[j] = x;
      }
    }
  __syncthreads();
  const int m = min(n, pre_max);              // order[:pre_maxsize]
  int kept = 0;
  for (int pos = 0; pos < m; ++pos) {
    const unsigned char st = state[pos];
    __syncthreads();
    if (st != 0) continue;
    if (tid == 0) state[pos] = 1;
    ++kept;
    const int i = (int)(0xffffffffu - (unsigned)(keys[pos] & 0xffffffffull));
    float bi[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) bi[k] = boxes[i * 5 + k];
    for (int q = pos + 1 + tid; q < m; q += NB_THREADS) {
      if (state[q] != 0) continue;
      const int j = (int)(0xffffffffu - (unsigned)(keys[q] & 0xffffffffull));
      float bj[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) bj[k] = boxes[j * 5 + k];
      if (ff3d_rot::iou_bev(bi, bj) > thresh) state[q] = 2;
    }
    __syncthreads();
  }
  __syncthreads();
  if (tid == 0) {
    int slot = 0;
    for (int pos = 0; pos < m && slot < post_max; ++pos)
      if (state[pos] == 1) keep[slot++] = (int)(0xffffffffu - (unsigned)(keys[pos] & 0xffffffffull));
    count[0] = slot;
  }
}

// 3-D IoU of LiDAR boxes (x, y, z_bottom, dx, dy, dz, yaw[, ...]) - mmdet3d 0.17.1 `BboxOverlaps3D(coordinate='lidar')`
// = `LiDARInstance3DBoxes.overlaps(mode='iou')` (un-vendored; restated): rotated BEV overlap AREA of xywhr2xyxyr(bev)
// (iou3d `boxes_overlap_bev_gpu`) times the height overlap max(0, min(top) - max(bottom)), over the union volume
// clamped at 1e-8.  The matching cost `IoU3DCost` of HungarianAssigner3D (hungarian_assigner.py:40-47, 128-129).
__global__ __launch_bounds__(256) void boxes_iou3d_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ out, int N, int M, int da, int db) {
  const long long total = (long long)N * M;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int i = (int)(e / M), j = (int)(e - (long long)i * M);
    const float* pa = a + (long long)i * da;
    const float* pb = b + (long long)j * db;
    const float ba[5] = {pa[0] - pa[3] / 2, pa[1] - pa[4] / 2, pa[0] + pa[3] / 2, pa[1] + pa[4] / 2, pa[6]};
    const float bb[5] = {pb[0] - pb[3] / 2, pb[1] - pb[4] / 2, pb[0] + pb[3] / 2, pb[1] + pb[4] / 2, pb[6]};
    const float oh = fmaxf(fminf(pa[2] + pa[5], pb[2] + pb[5]) - fmaxf(pa[2], pb[2]), 0.f);
    const float o3 = ff3d_rot::box_overlap(ba, bb) * oh;
    const float va = pa[3] * pa[4] * pa[5], vb = pb[3] * pb[4] * pb[5];
    out[e] = o3 / fmaxf(va + vb - o3, 1e-8f);
  }
}

}  // namespace

extern "C" int ff3d_boxes_iou_bev(const float* boxes_a, const float* boxes_b, float* out, int N, int M,
                                  ff3d_stream_t stream) {
  FF3D_REQUIRE(boxes_a && boxes_b && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(N > 0 && M > 0 && (long long)N * M < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  long long blocks = ((long long)N * M + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  ff3d_clear_error();
  hipLaunchKernelGGL(boxes_iou_bev_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     boxes_a, boxes_b, out, N, M);
  return ff3d_launch_status();
}

extern "C" int ff3d_nms_bev(const float* boxes, const float* scores, float thresh, int pre_max_size, int post_max_size,
                            int32_t* keep, int32_t* count, int n, ff3d_stream_t stream) {
  FF3D_REQUIRE(boxes && scores && keep && count, FF3D_ERR_NULL);
  FF3D_REQUIRE(n > 0 && n <= NB_MAX && pre_max_size > 0 && post_max_size > 0, FF3D_ERR_BAD_SHAPE);
  ff3d_clear_error();
  hipLaunchKernelGGL(nms_bev_kernel, dim3(1), dim3(NB_THREADS), 0, static_cast<hipStream_t>(stream), boxes, scores,
                     thresh, pre_max_size, post_max_size, keep, count, n);
  return ff3d_launch_status();
}

extern "C" int ff3d_boxes_iou3d(const float* boxes_a, const float* boxes_b, float* iou, int N, int M, int dim_a, int dim_b,
                                ff3d_stream_t stream) {
  FF3D_REQUIRE(boxes_a && boxes_b && iou, FF3D_ERR_NULL);
  FF3D_REQUIRE(N > 0 && M > 0 && dim_a >= 7 && dim_b >= 7, FF3D_ERR_BAD_SHAPE);
  long long blocks = ((long long)N * M + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  ff3d_clear_error();
  hipLaunchKernelGGL(boxes_iou3d_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), boxes_a,
                     boxes_b, iou, N, M, dim_a, dim_b);
  return ff3d_launch_status();
}
