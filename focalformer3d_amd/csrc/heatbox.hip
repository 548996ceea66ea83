// The heatmap_box branch of the Hard-Instance-Probing stages (reference FD:231-287 thin form, FD:708-722, FD:732-770; no shipped
// config enables it).  A (conv, conv) task head regresses one 10-value box per BEV cell and task group; the queries a stage selects
// start from the box of their (class, cell), and mask_heatmap_mode = 'boxcls' additionally blanks, for the next stages, every cell
// whose centre lies inside a selected query's (shrunk) box - in that query's class plane.
//
// Both kernels are index / byte work over a few MB: one thread per (frame, query) and one thread per BEV cell.  The point-in-box
// test follows mmdet3d v0.17.1's points_in_boxes_cuda.cu (un-vendored; call site FD:742,756-758): offset rotated by rz + pi / 2,
// strict inequalities, FIRST containing box wins.  oracle/ff3d_oracle.py: heatmap_box_gather, box_class_mask, points_in_boxes.
#include "ff3d_common.h"

namespace {

constexpr int HB_BOX = 10;        // reg 2, height 1, dim 3, rot 2, vel 2 (FD:240-243)
constexpr int HB_MAX_K = 32;
constexpr int HB_MAX_BOXES = 1024;

struct ClassTask {
  int t[HB_MAX_K];
};

// FD:708-722: box[(c, cell)] = task head output of the class's task group; x, y += the cell's integer coordinates
// (bev_pos.int()); clips of FD:714-717; gather at the stage's proposals.
__global__ __launch_bounds__(256) void heatmap_box_gather_kernel(const float* __restrict__ raw, const long long* __restrict__ idx,
                                                                 ClassTask ct, float* __restrict__ query_box, int B, int T, int H,
                                                                 int W, int k, int q_offset, int Nq) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * k) return;
  const int b = i / k, j = i - b * k;
  const int HW = H * W;
  const long long flat = idx[(long long)b * k + j];
  const int cls = (int)(flat / HW), cell = (int)(flat - (long long)cls * HW);
  const int y = cell / W, x = cell - y * W;
  const float* src = raw + ((long long)b * T + ct.t[cls]) * HB_BOX * HW + cell;
  const float lo_dim = (float)-0.6931471805599453, hi_dim = (float)2.70805020110221;     // np.log(0.5), np.log(15) (FD:715)
#pragma unroll
  for (int c = 0; c < HB_BOX; ++c) {
    float v = src[(long long)c * HW];
    if (c == 0) v += (float)x;
    else if (c == 1) v += (float)y;
    else if (c == 2) v = fminf(fmaxf(v, -5.f), 3.f);
    else if (c < 6) v = fminf(fmaxf(v, lo_dim), hi_dim);
    else if (c < 8) v = fminf(fmaxf(v, -1.f), 1.f);
    else v = fminf(fmaxf(v, -15.f), 15.f);
    query_box[((long long)b * HB_BOX + c) * Nq + q_offset + j] = v;
  }
}

struct BoxMaskParams {
  float osf, vx, vy, pcx, pcy;        // bbox coder (BC:46-60)
  float x0, y0, x1, y1;               // centre clip (FD:746-748)
  float margin, min_dim, max_dim;     // FD:749-753
};

// FD:732-768 + the dilation / accumulate of FD:774-782 for the box part of 'boxcls'.  Block = 256 cells of one frame; the frame's k
// decoded boxes sit in LDS.  A cell inside box j (first j) clears its 3 x 3 window (1 x 1 for the kernel-1 classes) in class plane
// label[j] of the accumulated mask: acc * (1 - maxpool(sel_pos OR sel_box)) = acc * (1 - maxpool(sel_pos)) * (1 - maxpool(sel_box))
// for {0,1} masks, and ff3d_query_gather has applied the first factor.
__global__ __launch_bounds__(256) void box_class_mask_kernel(const float* __restrict__ query_box, const long long* __restrict__ qlabel,
                                                             float* __restrict__ mask, int K, int H, int W, int k, int q_offset,
                                                             int Nq, BoxMaskParams p, int nms_kernel, uint32_t small_bits) {
  __shared__ float s_cx[HB_MAX_BOXES], s_cy[HB_MAX_BOXES], s_hw[HB_MAX_BOXES], s_hl[HB_MAX_BOXES], s_cos[HB_MAX_BOXES],
      s_sin[HB_MAX_BOXES];
  __shared__ int s_cls[HB_MAX_BOXES];
  const int b = blockIdx.y;
  const int HW = H * W;
  for (int j = threadIdx.x; j < k; j += 256) {
    const float* q = query_box + (long long)b * HB_BOX * Nq + q_offset + j;
    // decode_box (BC:54-69), float32 step by step as torch evaluates it
    float cx = __fadd_rn(__fmul_rn(__fmul_rn(q[0], p.osf), p.vx), p.pcx);
    float cy = __fadd_rn(__fmul_rn(__fmul_rn(q[(long long)Nq], p.osf), p.vy), p.pcy);
    cx = fminf(fmaxf(cx, p.x0), p.x1);
    cy = fminf(fmaxf(cy, p.y0), p.y1);
    float w = expf(q[3ll * Nq]), l = expf(q[4ll * Nq]);
    w = fminf(fmaxf(__fsub_rn(w, p.margin), p.min_dim), p.max_dim);
    l = fminf(fmaxf(__fsub_rn(l, p.margin), p.min_dim), p.max_dim);
    const float rz = atan2f(q[6ll * Nq], q[7ll * Nq]);
    const float rot = (float)((double)rz + 1.57079632679489661923);        // `float rot_angle = rz + M_PI / 2`
    s_cx[j] = cx, s_cy[j] = cy, s_hw[j] = w * 0.5f, s_hl[j] = l * 0.5f, s_cos[j] = cosf(rot), s_sin[j] = sinf(rot);
    s_cls[j] = (int)qlabel[(long long)b * Nq + q_offset + j];
  }
  __syncthreads();
  const int cell = blockIdx.x * 256 + threadIdx.x;
  if (cell >= HW) return;
  const int y = cell / W, x = cell - y * W;
  // decode_center of the cell centre (x + 0.5, y + 0.5) (BC:46-52); z = 0 always lies inside the box's z extent [-100, 900] (FD:754-755)
  const float px = __fadd_rn(__fmul_rn(__fmul_rn((float)x + 0.5f, p.osf), p.vx), p.pcx);
  const float py = __fadd_rn(__fmul_rn(__fmul_rn((float)y + 0.5f, p.osf), p.vy), p.pcy);
  int cls = -1;
  for (int j = 0; j < k; ++j) {
    const float sx = px - s_cx[j], sy = py - s_cy[j];
    const float lx = sx * s_cos[j] + sy * (-s_sin[j]);
    const float ly = sx * s_sin[j] + sy * s_cos[j];
    if (lx > -s_hl[j] && lx < s_hl[j] && ly > -s_hw[j] && ly < s_hw[j]) {
      cls = s_cls[j];
      break;
    }
  }
  if (cls < 0 || cls >= K) return;
  const bool small = ((small_bits >> cls) & 1u) || nms_kernel == 1;
  float* plane = mask + ((long long)b * K + cls) * HW;
  if (small) {
    plane[cell] = 0.f;
    return;
  }
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) plane[yy * W + xx] = 0.f;
    }
}

}  // namespace

extern "C" int ff3d_heatmap_box_gather(const float* raw, const int64_t* idx, const int32_t* class_task_host, float* query_box,
                                       int B, int K, int T, int H, int W, int k, int q_offset, int Nq, ff3d_stream_t stream) {
  FF3D_REQUIRE(raw && idx && class_task_host && query_box, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && K > 0 && K <= HB_MAX_K && T > 0 && H > 0 && W > 0 && k > 0 && q_offset >= 0 && q_offset + k <= Nq,
               FF3D_ERR_BAD_SHAPE);
  ClassTask ct;
  for (int c = 0; c < HB_MAX_K; ++c) ct.t[c] = 0;
  for (int c = 0; c < K; ++c) {
    FF3D_REQUIRE(class_task_host[c] >= 0 && class_task_host[c] < T, FF3D_ERR_BAD_SHAPE);
    ct.t[c] = class_task_host[c];
  }
  ff3d_clear_error();
  hipLaunchKernelGGL(heatmap_box_gather_kernel, dim3((B * k + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), raw,
                     reinterpret_cast<const long long*>(idx), ct, query_box, B, T, H, W, k, q_offset, Nq);
  return ff3d_launch_status();
}

extern "C" int ff3d_box_class_mask(const float* query_box, const int64_t* qlabel, float* mask, int B, int K, int H, int W, int k,
                                   int q_offset, int Nq, const float* coder_host, const float* range_host, float margin,
                                   float min_bev_dim, float max_bev_dim, int nms_kernel, uint32_t small_class_bits,
                                   ff3d_stream_t stream) {
  FF3D_REQUIRE(query_box && qlabel && mask && coder_host && range_host, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && B <= 65535 && K > 0 && K <= HB_MAX_K && H > 0 && W > 0 && k > 0 && k <= HB_MAX_BOXES && q_offset >= 0 &&
                   q_offset + k <= Nq,
               FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(nms_kernel == 1 || nms_kernel == 3, FF3D_ERR_UNSUPPORTED);
  BoxMaskParams p{coder_host[0], coder_host[1], coder_host[2], coder_host[3], coder_host[4], range_host[0], range_host[1],
                  range_host[2], range_host[3], margin,        min_bev_dim,   max_bev_dim};
  ff3d_clear_error();
  hipLaunchKernelGGL(box_class_mask_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, static_cast<hipStream_t>(stream),
                     query_box, reinterpret_cast<const long long*>(qlabel), mask, K, H, W, k, q_offset, Nq, p, nms_kernel,
                     small_class_bits);
  return ff3d_launch_status();
}
