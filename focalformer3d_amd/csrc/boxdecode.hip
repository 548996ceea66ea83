// get_bboxes for gfx950: score fusion, box decode, range filter, order-preserving compaction and
// the 200-box cap, one block per frame, no host round trip (the reference compacts with boolean
// masks -> a device sync per frame).  Elementwise + one LDS bitonic sort; HBM-trivial.
#include "ff3d_common.h"

namespace {

constexpr int BD_THREADS = 256;
constexpr int BD_MAXQ = 4096;

struct BoxParams {
  const float *cls, *center, *height, *dim, *rot, *vel, *qscore;
  const long long* qlabel;
  float *boxes, *scores;
  int* labels;
  int* count;
  long long ld;
  int q0, K, Nq, max_out, box_dim;
  float osf, vx, vy, pcx, pcy;
  float lo[3], hi[3];
  float thr;
  int no_filter;     // BC:71-158 decode(filter=False): every query kept, in query order - NaN / inf boxes included
};

struct Decoded {
  float v[9];
  float score;
  int label;
  bool keep;
};

__device__ __forceinline__ Decoded decode_one(const BoxParams& p, int b, int q) {
  Decoded d;
  const long long col = p.q0 + q;
  const int lab = (int)p.qlabel[(long long)b * p.Nq + q];
  // FD:1317-1321: sigmoid(cls) * query_heatmap_score * one_hot(label); only column `lab` survives
  // the one-hot, so max/argmax over classes (BC:86-87) reduce to that entry (argmax of an all-zero
  // column is implementation-defined in the reference; index 0 here, as torch.max returns).
  const float logit = p.cls[((long long)b * p.K + lab) * p.ld + col];
  const float s = (1.f / (1.f + expf(-logit))) * p.qscore[((long long)b * p.K + lab) * p.Nq + q];
  d.score = s;
  d.label = s > 0.f ? lab : 0;
  const float x = p.center[((long long)b * 2 + 0) * p.ld + col] * p.osf * p.vx + p.pcx;  // BC:90-91
  const float y = p.center[((long long)b * 2 + 1) * p.ld + col] * p.osf * p.vy + p.pcy;
  const float w = expf(p.dim[((long long)b * 3 + 0) * p.ld + col]);
  const float l = expf(p.dim[((long long)b * 3 + 1) * p.ld + col]);
  const float h = expf(p.dim[((long long)b * 3 + 2) * p.ld + col]);
  const float z = p.height[(long long)b * p.ld + col] - h * 0.5f;                           // BC:96
  const float yaw = atan2f(p.rot[((long long)b * 2 + 0) * p.ld + col], p.rot[((long long)b * 2 + 1) * p.ld + col]);
  d.v[0] = x; d.v[1] = y; d.v[2] = z; d.v[3] = w; d.v[4] = l; d.v[5] = h; d.v[6] = yaw;
  d.v[7] = d.v[8] = 0.f;
  if (p.vel) {
    d.v[7] = p.vel[((long long)b * 2 + 0) * p.ld + col];
    d.v[8] = p.vel[((long long)b * 2 + 1) * p.ld + col];
  }
  bool keep = x >= p.lo[0] && y >= p.lo[1] && z >= p.lo[2] && x <= p.hi[0] && y <= p.hi[1] && z <= p.hi[2];
  if (p.thr != 0.f) keep = keep && (s > p.thr);  // BC:140-141: applied only when the threshold is truthy
  d.keep = keep || p.no_filter;
  return d;
}

__device__ __forceinline__ void write_row(const BoxParams& p, int b, int slot, const Decoded& d) {
  float* o = p.boxes + ((long long)b * p.max_out + slot) * p.box_dim;
  for (int i = 0; i < p.box_dim; ++i) o[i] = d.v[i];
  p.scores[(long long)b * p.max_out + slot] = d.score;
  p.labels[(long long)b * p.max_out + slot] = d.label;
}

__global__ __launch_bounds__(BD_THREADS) void box_decode_kernel(BoxParams p) {
  __shared__ unsigned long long keys[BD_MAXQ];
  __shared__ int s_scan[BD_THREADS];
  __shared__ int s_total;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int per = (p.Nq + BD_THREADS - 1) / BD_THREADS;  // consecutive queries per thread (order-preserving)
  const int qa = tid * per, qb = min(qa + per, p.Nq);
  if (p.no_filter) {                                     // slot = query (max_out >= Nq checked by the host side)
    for (int q = qa; q < qb; ++q) write_row(p, b, q, decode_one(p, b, q));
    if (tid == 0) p.count[b] = p.Nq;
    return;
  }

  int mine = 0;
  for (int q = qa; q < qb; ++q) mine += decode_one(p, b, q).keep ? 1 : 0;
  s_scan[tid] = mine;
  __syncthreads();
  for (int off = 1; off < BD_THREADS; off <<= 1) {  // inclusive prefix scan
    const int add = tid >= off ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += add;
    __syncthreads();
  }
  const int total = s_scan[BD_THREADS - 1];
  int slot = s_scan[tid] - mine;

  if (total <= p.max_out) {
    for (int q = qa; q < qb; ++q) {
      const Decoded d = decode_one(p, b, q);
      if (d.keep) write_row(p, b, slot++, d);
    }
    if (tid == 0) p.count[b] = total;
    return;  // uniform branch: no barrier below is skipped by a subset of threads
  }
  // FD:1395-1400: more than max_out (=200) boxes -> best max_out by score, descending
  for (int q = qa; q < qb; ++q) {
    const Decoded d = decode_one(p, b, q);
    if (d.keep) keys[slot++] = ((unsigned long long)__float_as_uint(fmaxf(d.score, 0.f)) << 32) |
                               (unsigned long long)(0xffffffffu - (unsigned)q);
  }
  int n2 = 2;
  while (n2 < total) n2 <<= 1;
  __syncthreads();
  for (int i = total + tid; i < n2; i += BD_THREADS) keys[i] = 0ull;
  for (int size = 2; size <= n2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = tid; i < (n2 >> 1); i += BD_THREADS) {
        const int pos = 2 * i - (i & (stride - 1)), j = pos + stride;
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
  for (int j = tid; j < p.max_out; j += BD_THREADS) {
    const int q = (int)(0xffffffffu - (unsigned)(keys[j] & 0xffffffffull));
    write_row(p, b, j, decode_one(p, b, q));
  }
  if (tid == 0) p.count[b] = p.max_out;
}

}  // namespace

extern "C" int ff3d_box_decode(const float* cls, const float* center, const float* height, const float* dim,
                               const float* rot, const float* vel, int64_t ld, int q0, const float* qscore,
                               const int64_t* qlabel, float* boxes, float* scores, int32_t* labels, int32_t* count,
                               int B, int K, int Nq, int max_out, const float* coder_host,
                               const float* post_center_range_host, float score_threshold, ff3d_stream_t stream) {
  FF3D_REQUIRE(cls && center && height && dim && rot && qscore && qlabel && boxes && scores && labels && count && coder_host,
               FF3D_ERR_NULL);
  const bool no_filter = post_center_range_host == nullptr;      // decode(filter=False): needs room for every query
  FF3D_REQUIRE(B > 0 && K > 0 && Nq > 0 && max_out > 0 && q0 >= 0 && q0 + Nq <= ld, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(no_filter ? max_out >= Nq : Nq <= BD_MAXQ, FF3D_ERR_BAD_SHAPE);
  BoxParams p;
  p.cls = cls; p.center = center; p.height = height; p.dim = dim; p.rot = rot; p.vel = vel; p.qscore = qscore;
  p.qlabel = reinterpret_cast<const long long*>(qlabel);
  p.boxes = boxes; p.scores = scores; p.labels = labels; p.count = count;
  p.ld = ld; p.q0 = q0; p.K = K; p.Nq = Nq; p.max_out = max_out; p.box_dim = vel ? 9 : 7;
  p.osf = coder_host[0]; p.vx = coder_host[1]; p.vy = coder_host[2]; p.pcx = coder_host[3]; p.pcy = coder_host[4];
  for (int i = 0; i < 3; ++i) {
    p.lo[i] = no_filter ? 0.f : post_center_range_host[i];
    p.hi[i] = no_filter ? 0.f : post_center_range_host[3 + i];
  }
  p.thr = score_threshold;
  p.no_filter = no_filter ? 1 : 0;
  ff3d_clear_error();
  hipLaunchKernelGGL(box_decode_kernel, dim3(B), dim3(BD_THREADS), 0, static_cast<hipStream_t>(stream), p);
  return ff3d_launch_status();
}

// ---- packed detections for the multi-GPU gather (dist.py): (B, M+1, 11) fp32, row 0 = (count, box_dim, 0...),
// rows 1.. = box values zero-padded to 9 | score | label.  One launch instead of five framework ops per batch.
namespace {
__global__ __launch_bounds__(256) void pack_detections_kernel(const float* __restrict__ boxes,
                                                              const float* __restrict__ scores,
                                                              const int32_t* __restrict__ labels,
                                                              const int32_t* __restrict__ count, float* __restrict__ out,
                                                              int M, int D, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % 11);
    const long long r = i / 11;
    const int row = (int)(r % (M + 1)), b = (int)(r / (M + 1));
    float v = 0.f;
    if (row == 0) {
      v = c == 0 ? (float)count[b] : c == 1 ? (float)D : 0.f;
    } else {
      const long long j = (long long)b * M + (row - 1);
      v = c < D ? boxes[j * D + c] : c == 9 ? scores[j] : c == 10 ? (float)labels[j] : 0.f;
    }
    out[i] = v;
  }
}
}  // namespace

extern "C" int ff3d_pack_detections(const float* boxes, const float* scores, const int32_t* labels, const int32_t* count,
                                    float* packed, int B, int M, int box_dim, ff3d_stream_t stream) {
  FF3D_REQUIRE(boxes && scores && labels && count && packed, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && M > 0 && box_dim > 0 && box_dim <= 9 && M < (1 << 24), FF3D_ERR_BAD_SHAPE);
  const long long total = (long long)B * (M + 1) * 11;
  long long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  ff3d_clear_error();
  hipLaunchKernelGGL(pack_detections_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), boxes,
                     scores, labels, count, packed, M, box_dim, total);
  return ff3d_launch_status();
}
