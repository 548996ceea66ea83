// Rotated BEV IoU of two boxes given as (x1, y1, x2, y2, angle) - device functions shared by the NMS kernels.
//
// Counterpart of mmdet3d 0.17.1 `mmdet3d/ops/iou3d/src/iou3d_kernel.cu` (`box_overlap`, `iou_bev`; un-vendored third
// party, restated from its published algorithm): rotate the 4 corners of both rectangles about their
// centres, collect the proper edge-edge intersection points and the corners of either box lying inside the other
// (1e-5 margin), order the points by angle about their mean, fan-triangulate.  fp32 throughout, EPS = 1e-8.
#pragma once
#include <hip/hip_runtime.h>

namespace ff3d_rot {

struct P2 {
  float x, y;
};

__device__ __forceinline__ float cross2(P2 a, P2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ float cross3(P2 p1, P2 p2, P2 p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__device__ __forceinline__ bool rect_cross(P2 p1, P2 p2, P2 q1, P2 q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

__device__ __forceinline__ P2 rotate_about(P2 c, float cs, float sn, P2 p) {
  return P2{(p.x - c.x) * cs + (p.y - c.y) * sn + c.x, -(p.x - c.x) * sn + (p.y - c.y) * cs + c.y};
}

// point inside the (rotated) box, with the reference's 1e-5 margin; cs/sn = cos/sin of -angle
__device__ __forceinline__ bool in_box(const float* box, float cs_neg, float sn_neg, P2 p) {
  const float MARGIN = 1e-5f;
  const P2 c{(box[0] + box[2]) / 2, (box[1] + box[3]) / 2};
  const P2 r = rotate_about(c, cs_neg, sn_neg, p);
  return r.x > box[0] - MARGIN && r.x < box[2] + MARGIN && r.y > box[1] - MARGIN && r.y < box[3] + MARGIN;
}

__device__ __forceinline__ bool seg_intersection(P2 p1, P2 p0, P2 q1, P2 q0, P2& ans) {
  const float EPS = 1e-8f;
  if (!rect_cross(p0, p1, q0, q1)) return false;
  const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > EPS) {
    ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans.x = (b0 * c1 - b1 * c0) / D;
    ans.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

__device__ inline float box_overlap(const float* a, const float* b) {
  const P2 ca{(a[0] + a[2]) / 2, (a[1] + a[3]) / 2}, cb{(b[0] + b[2]) / 2, (b[1] + b[3]) / 2};
  P2 A[5] = {{a[0], a[1]}, {a[2], a[1]}, {a[2], a[3]}, {a[0], a[3]}, {0, 0}};
  P2 Bc[5] = {{b[0], b[1]}, {b[2], b[1]}, {b[2], b[3]}, {b[0], b[3]}, {0, 0}};
  const float acs = cosf(a[4]), asn = sinf(a[4]), bcs = cosf(b[4]), bsn = sinf(b[4]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    A[k] = rotate_about(ca, acs, asn, A[k]);
    Bc[k] = rotate_about(cb, bcs, bsn, Bc[k]);
  }
  A[4] = A[0];
  Bc[4] = Bc[0];
  P2 pts[24];
  P2 centre{0.f, 0.f};
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      P2 x;
      if (seg_intersection(A[i + 1], A[i], Bc[j + 1], Bc[j], x)) {
        centre.x += x.x, centre.y += x.y;
        pts[cnt++] = x;
      }
    }
  const float acn = cosf(-a[4]), asnn = sinf(-a[4]), bcn = cosf(-b[4]), bsnn = sinf(-b[4]);
  for (int k = 0; k < 4; ++k) {
    if (in_box(a, acn, asnn, Bc[k])) {
      centre.x += Bc[k].x, centre.y += Bc[k].y;
      pts[cnt++] = Bc[k];
    }
    if (in_box(b, bcn, bsnn, A[k])) {
      centre.x += A[k].x, centre.y += A[k].y;
      pts[cnt++] = A[k];
    }
  }
  if (cnt < 3) return 0.f;                    // no polygon (the reference's fan sum is empty / degenerate here as well)
  centre.x /= cnt, centre.y /= cnt;
  float ang[24];
  for (int k = 0; k < cnt; ++k) ang[k] = atan2f(pts[k].y - centre.y, pts[k].x - centre.x);
  for (int j = 0; j < cnt - 1; ++j)           // the reference's bubble sort (ascending polar angle)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (ang[i] > ang[i + 1]) {
        const P2 t = pts[i];
        pts[i] = pts[i + 1], pts[i + 1] = t;
        const float ta = ang[i];
        ang[i] = ang[i + 1], ang[i + 1] = ta;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k)
    area += cross2(P2{pts[k].x - pts[0].x, pts[k].y - pts[0].y}, P2{pts[k + 1].x - pts[0].x, pts[k + 1].y - pts[0].y});
  return fabsf(area) / 2.0f;
}

__device__ inline float iou_bev(const float* a, const float* b) {
  const float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
  const float so = box_overlap(a, b);
  return so / fmaxf(sa + sb - so, 1e-8f);
}

}  // namespace ff3d_rot
