// Per-task circle NMS of get_bboxes (FD:1352-1393 with test_cfg.nms_type == 'circle') for gfx950, fused with the
// keep-mask compaction and the 200-box cap (FD:1387-1392): one block per frame, everything in LDS, no host round trip
// (the reference copies every task's boxes to the host and runs mmdet3d's numba `circle_nms` there, FD:1361-1367).
//
// mmdet3d 0.17.1 `circle_nms(dets[x, y, score], thresh, post_max_size=83)` (un-vendored third party; restated from its
// published algorithm): visit boxes by descending score; an unsuppressed box is kept and suppresses every later box of
// the same task whose squared centre distance is <= thresh; the first post_max_size kept boxes of the task survive.
// Tasks with radius <= 0 keep all their boxes (FD:1378-1379).  Suppression never crosses tasks, so ONE sweep over
// the score-sorted list serves all tasks: the sweep is sequential (n <= 4096 steps), the suppression of each kept
// box is parallel over the block.
//
// The same sweep serves test_cfg.nms_type == 'rotate' (FD:1369-1377, mmdet3d 0.17.1 `nms_gpu` on the xyxyr BEV boxes:
// sort by score, keep the task's first pre_maxsize, suppress later boxes whose rotated BEV IoU is > thresh, keep the
// first post_max_size survivors) - kernel template parameter ROT; the IoU is rotiou.h.
#include "ff3d_common.h"
#include "rotiou.h"

namespace {

constexpr int CN_THREADS = 256, CN_MAX = 2048, CN_MAX_TASKS = 16;

struct CircleParams {
  const float *boxes, *scores;
  const int *labels, *count;
  float *out_boxes, *out_scores;
  int *out_labels, *out_count;
  int M, box_dim, max_out, post_max, pre_max, num_tasks, K;
  int class_task[32];
  float radius[CN_MAX_TASKS];
};

__device__ void sort_desc(unsigned long long* keys, int n2) {
  for (int size = 2; size <= n2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < (n2 >> 1); i += CN_THREADS) {
        const int pos = 2 * i - (i & (stride - 1)), j = pos + stride;
        const bool up = (pos & size) == 0;
        const unsigned long long a = keys[pos], c = keys[j];
        if ((a < c) == up) {
          keys[pos] = c;
          keys[j] = a;
        }
      }
    }
  __syncthreads();
}

template <bool ROT>
__global__ __launch_bounds__(CN_THREADS) void circle_nms_kernel(CircleParams p) {
  __shared__ unsigned long long keys[CN_MAX];
  __shared__ float sx[CN_MAX], sy[CN_MAX];
  __shared__ float sgeo[ROT ? 3 : 1][ROT ? CN_MAX : 1];      // ROT: x2, y2, angle (sx, sy hold x1, y1)
  __shared__ unsigned char stask[CN_MAX], sstate[CN_MAX];   // state: 0 undecided, 1 kept, 2 suppressed / dropped
  __shared__ int s_scan[CN_THREADS];
  __shared__ int task_kept[CN_MAX_TASKS];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = min(p.count[b], min(p.M, CN_MAX));
  const float* bx = p.boxes + (long long)b * p.M * p.box_dim;
  const float* sc = p.scores + (long long)b * p.M;
  const int* lb = p.labels + (long long)b * p.M;

  int n2 = 2;
  while (n2 < n) n2 <<= 1;
  for (int i = tid; i < n2; i += CN_THREADS) {
    if (i < n) {
      keys[i] = ((unsigned long long)__float_as_uint(fmaxf(sc[i], 0.f)) << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
      const float* bi = bx + (long long)i * p.box_dim;
      if (ROT) {                               // .bev = (x, y, w, l, yaw) -> xywhr2xyxyr (FD:1369)
        const float hw = bi[3] / 2, hl = bi[4] / 2;
        sx[i] = bi[0] - hw, sy[i] = bi[1] - hl;
        sgeo[0][i] = bi[0] + hw, sgeo[1][i] = bi[1] + hl, sgeo[2][i] = bi[6];
      } else {
        sx[i] = bi[0], sy[i] = bi[1];
      }
      const int l = lb[i];
      stask[i] = (l >= 0 && l < p.K) ? (unsigned char)p.class_task[l] : 255;
      sstate[i] = 0;
    } else {
      keys[i] = 0ull;
    }
  }
  if (tid < CN_MAX_TASKS) task_kept[tid] = 0;
  sort_desc(keys, n2);   // score descending (ties: lower index first)
  // nms_gpu's order[:pre_maxsize]: later boxes of the task are dropped outright - only where nms_gpu runs: a task with
  // radius <= 0 keeps ALL its boxes (FD:1378-1379; found by the reference-executed fixture get_bboxes_nms_nuscenes.npz, whose
  // no-NMS task is larger than pre_maxsize)
  if (ROT && tid < p.num_tasks && p.radius[tid] > 0.f) {
    int seen = 0;
    for (int pos = 0; pos < n; ++pos) {
      const int i = (int)(0xffffffffu - (unsigned)(keys[pos] & 0xffffffffull));
      if (stask[i] == tid && seen++ >= p.pre_max) sstate[i] = 2;
    }
  }
  __syncthreads();

  // ---- sweep in score order
  for (int pos = 0; pos < n; ++pos) {
    const int i = (int)(0xffffffffu - (unsigned)(keys[pos] & 0xffffffffull));
    const int t = stask[i];
    const unsigned char st = sstate[i];       // uniform read
    __syncthreads();
    if (t >= p.num_tasks) {                   // class outside every task: never kept (FD:1353-1358 task masks)
      if (tid == 0) sstate[i] = 2;
      continue;
    }
    if (st != 0) continue;
    const float r = p.radius[t];
    if (tid == 0) {
      sstate[i] = (r > 0.f && task_kept[t] >= p.post_max) ? 2 : 1;   // keep[:post_max_size]
      task_kept[t] += 1;
    }
    if (r > 0.f) {
      const float xi = sx[i], yi = sy[i];
      float bi[5] = {xi, yi, 0.f, 0.f, 0.f};
      if (ROT) bi[2] = sgeo[0][i], bi[3] = sgeo[1][i], bi[4] = sgeo[2][i];
      for (int q = pos + 1 + tid; q < n; q += CN_THREADS) {
        const int jx = (int)(0xffffffffu - (unsigned)(keys[q] & 0xffffffffull));
        if (stask[jx] == t && sstate[jx] == 0) {
          if (ROT) {
            const float bj[5] = {sx[jx], sy[jx], sgeo[0][jx], sgeo[1][jx], sgeo[2][jx]};
            if (ff3d_rot::iou_bev(bi, bj) > r) sstate[jx] = 2;
          } else {
            const float dx = xi - sx[jx], dy = yi - sy[jx];
            if (dx * dx + dy * dy <= r) sstate[jx] = 2;
          }
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();

  // ---- compaction in ORIGINAL order (boxes3d[keep_mask], FD:1383-1386), then the cap by score (FD:1387-1392)
  const int per = (n + CN_THREADS - 1) / CN_THREADS;
  const int qa = min(tid * per, n), qb = min(qa + per, n);
  int mine = 0;
  for (int i = qa; i < qb; ++i) mine += sstate[i] == 1;
  s_scan[tid] = mine;
  __syncthreads();
  for (int off = 1; off < CN_THREADS; off <<= 1) {
    const int add = tid >= off ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += add;
    __syncthreads();
  }
  const int total = s_scan[CN_THREADS - 1];
  auto emit = [&](int slot, int i) {
    float* o = p.out_boxes + ((long long)b * p.max_out + slot) * p.box_dim;
    for (int d = 0; d < p.box_dim; ++d) o[d] = bx[(long long)i * p.box_dim + d];
    p.out_scores[(long long)b * p.max_out + slot] = sc[i];
    p.out_labels[(long long)b * p.max_out + slot] = lb[i];
  };
  if (total <= p.max_out) {
    int slot = s_scan[tid] - mine;
    for (int i = qa; i < qb; ++i)
      if (sstate[i] == 1) emit(slot++, i);
    if (tid == 0) p.out_count[b] = total;
    return;
  }
  // more than max_out survivors: the sorted key list already orders them by score
  __syncthreads();
  if (tid == 0) {
    int slot = 0;
    for (int pos = 0; pos < n && slot < p.max_out; ++pos) {
      const int i = (int)(0xffffffffu - (unsigned)(keys[pos] & 0xffffffffull));
      if (sstate[i] == 1) emit(slot++, i);
    }
    p.out_count[b] = p.max_out;
  }
}

}  // namespace

static int nms_launch(bool rot, const float* boxes, const float* scores, const int32_t* labels, const int32_t* count,
                      float* out_boxes, float* out_scores, int32_t* out_labels, int32_t* out_count, int B, int M,
                      int box_dim, int max_out, int K, const int32_t* class_task_host, int num_tasks,
                      const float* task_thresh_host, int pre_max_size, int post_max_size, ff3d_stream_t stream) {
  FF3D_REQUIRE(boxes && scores && labels && count && out_boxes && out_scores && out_labels && out_count &&
                   class_task_host && task_thresh_host,
               FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && M > 0 && M <= CN_MAX && box_dim >= (rot ? 7 : 2) && max_out > 0 && K > 0 && K <= 32 &&
                   num_tasks > 0 && num_tasks <= CN_MAX_TASKS && post_max_size > 0 && pre_max_size > 0,
               FF3D_ERR_BAD_SHAPE);
  CircleParams p;
  p.boxes = boxes; p.scores = scores; p.labels = labels; p.count = count;
  p.out_boxes = out_boxes; p.out_scores = out_scores; p.out_labels = out_labels; p.out_count = out_count;
  p.M = M; p.box_dim = box_dim; p.max_out = max_out; p.post_max = post_max_size; p.pre_max = pre_max_size;
  p.num_tasks = num_tasks; p.K = K;
  for (int i = 0; i < 32; ++i) p.class_task[i] = i < K ? class_task_host[i] : 255;
  for (int i = 0; i < CN_MAX_TASKS; ++i) p.radius[i] = i < num_tasks ? task_thresh_host[i] : 0.f;
  ff3d_clear_error();
  if (rot)
    hipLaunchKernelGGL(circle_nms_kernel<true>, dim3(B), dim3(CN_THREADS), 0, static_cast<hipStream_t>(stream), p);
  else
    hipLaunchKernelGGL(circle_nms_kernel<false>, dim3(B), dim3(CN_THREADS), 0, static_cast<hipStream_t>(stream), p);
  return ff3d_launch_status();
}

extern "C" int ff3d_circle_nms(const float* boxes, const float* scores, const int32_t* labels, const int32_t* count,
                               float* out_boxes, float* out_scores, int32_t* out_labels, int32_t* out_count, int B,
                               int M, int box_dim, int max_out, int K, const int32_t* class_task_host, int num_tasks,
                               const float* task_radius_host, int post_max_size, ff3d_stream_t stream) {
  return nms_launch(false, boxes, scores, labels, count, out_boxes, out_scores, out_labels, out_count, B, M, box_dim,
                    max_out, K, class_task_host, num_tasks, task_radius_host, 1 << 30, post_max_size, stream);
}

extern "C" int ff3d_rotate_nms(const float* boxes, const float* scores, const int32_t* labels, const int32_t* count,
                               float* out_boxes, float* out_scores, int32_t* out_labels, int32_t* out_count, int B,
                               int M, int box_dim, int max_out, int K, const int32_t* class_task_host, int num_tasks,
                               const float* task_thresh_host, int pre_max_size, int post_max_size,
                               ff3d_stream_t stream) {
  return nms_launch(true, boxes, scores, labels, count, out_boxes, out_scores, out_labels, out_count, B, M, box_dim,
                    max_out, K, class_task_host, num_tasks, task_thresh_host, pre_max_size, post_max_size, stream);
}
