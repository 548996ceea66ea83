// Camera-projection feature sampler (I2P) for gfx950.
//
// The reference materialises (Ncam, C, Z*H*W) sampled features (~1 GB at the BASELINE shape), reduces
// over cameras, then runs a 1-head attention over the Z height samples of each BEV pillar.  Here one
// group of Ci/4 lanes owns one pillar: for every height sample it projects the point into each
// camera, bilinear-samples the NHWC camera map (each corner = Ci contiguous floats, 16-byte loads),
// takes the masked multi-view mean, and feeds an online softmax (running max / sum / weighted
// accumulator in registers) with the folded query qk - the (HW, Z, C) key/value tensor never exists.
// The dot product qk.f_z is reduced across the lane group with DPP/shuffle butterflies.
// HBM-bound gather; dense q/k/v/out projections stay outside as GEMMs.
#include "ff3d_common.h"

namespace {

struct CamParams {
  const float *img_cl, *lidar2img, *img_aug, *qk;
  float* ctx;
  uint8_t* valid;
  int Ncam, Ci, Hi, Wi, H, W, Z;
  float lo[3], hi[3];
  float in_h, in_w;
};

template <int LPG>  // lanes per pillar = Ci/4 rounded up to a power of two (<= 64), extra lanes idle
__global__ __launch_bounds__(256) void cam_sample_kernel(CamParams p, int C4) {
  constexpr int PPB = 256 / LPG;
  const int tid = threadIdx.x, pl = tid / LPG, sub = tid - pl * LPG;
  const long long pillar = (long long)blockIdx.x * PPB + pl;
  const int HW = p.H * p.W;
  const int b = blockIdx.y;
  const bool active = pillar < HW;
  const int cell = active ? (int)pillar : 0;
  const int yy = cell / p.W, xx = cell - yy * p.W;
  const bool lane_on = sub < C4;
  const int coff = lane_on ? sub * 4 : 0;

  float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active && lane_on) q4 = *reinterpret_cast<const float4*>(p.qk + ((long long)b * HW + cell) * p.Ci + coff);

  // pillar point in metric lidar coordinates (EU:210-214): ((idx + 0.5) / size) * range + min
  const float px = ((float)xx + 0.5f) / (float)p.W * (p.hi[0] - p.lo[0]) + p.lo[0];
  const float py = ((float)yy + 0.5f) / (float)p.H * (p.hi[1] - p.lo[1]) + p.lo[1];

  float m_run = -INFINITY, l_run = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  bool any = false;

  for (int z = 0; z < p.Z; ++z) {
    const float pz = ((float)z + 0.5f) / (float)p.Z * (p.hi[2] - p.lo[2]) + p.lo[2];
    float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
    float cnt = 0.f;
    for (int cam = 0; cam < p.Ncam; ++cam) {
      const float* M = p.lidar2img + ((long long)b * p.Ncam + cam) * 16;
      const float cx = M[0] * px + M[1] * py + M[2] * pz + M[3];
      const float cy = M[4] * px + M[5] * py + M[6] * pz + M[7];
      const float cz = M[8] * px + M[9] * py + M[10] * pz + M[11];
      bool ok = cz > 1e-5f;                                               // EU:226-227
      const float den = fmaxf(cz, 1e-5f);
      float u = cx / den, v = cy / den;
      if (p.img_aug) {                                                    // EU:230-233
        const float* A = p.img_aug + ((long long)b * p.Ncam + cam) * 16;
        const float u2 = A[0] * u + A[1] * v + A[2] + A[3];
        const float v2 = A[4] * u + A[5] * v + A[6] + A[7];
        u = u2;
        v = v2;
      }
      u = (u / p.in_w - 0.5f) * 2.f;                                      // EU:234-237
      v = (v / p.in_h - 0.5f) * 2.f;
      ok = ok && u > -1.f && u < 1.f && v > -1.f && v < 1.f;              // EU:238-241
      if (!ok) continue;                                                  // uniform within the lane group
      cnt += 1.f;
      // F.grid_sample bilinear, zeros padding, align_corners=False
      const float ix = ((u + 1.f) * (float)p.Wi - 1.f) / 2.f, iy = ((v + 1.f) * (float)p.Hi - 1.f) / 2.f;
      const float x0f = floorf(ix), y0f = floorf(iy);
      const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
      const float lx = ix - x0f, ly = iy - y0f, hx = 1.f - lx, hy = 1.f - ly;
      const bool vx0 = (unsigned)x0 < (unsigned)p.Wi, vx1 = (unsigned)x1 < (unsigned)p.Wi;
      const bool vy0 = (unsigned)y0 < (unsigned)p.Hi, vy1 = (unsigned)y1 < (unsigned)p.Hi;
      const float w00 = (vy0 && vx0) ? hy * hx : 0.f, w01 = (vy0 && vx1) ? hy * lx : 0.f;
      const float w10 = (vy1 && vx0) ? ly * hx : 0.f, w11 = (vy1 && vx1) ? ly * lx : 0.f;
      const int cx0 = min(max(x0, 0), p.Wi - 1), cx1 = min(max(x1, 0), p.Wi - 1);
      const int cy0 = min(max(y0, 0), p.Hi - 1), cy1 = min(max(y1, 0), p.Hi - 1);
      const float* base = p.img_cl + (((long long)b * p.Ncam + cam) * p.Hi * p.Wi) * p.Ci + coff;
      if (active && lane_on) {
        const float4 a = *reinterpret_cast<const float4*>(base + (long long)(cy0 * p.Wi + cx0) * p.Ci);
        const float4 bq = *reinterpret_cast<const float4*>(base + (long long)(cy0 * p.Wi + cx1) * p.Ci);
        const float4 c = *reinterpret_cast<const float4*>(base + (long long)(cy1 * p.Wi + cx0) * p.Ci);
        const float4 d = *reinterpret_cast<const float4*>(base + (long long)(cy1 * p.Wi + cx1) * p.Ci);
        f.x += a.x * w00 + bq.x * w01 + c.x * w10 + d.x * w11;
        f.y += a.y * w00 + bq.y * w01 + c.y * w10 + d.y * w11;
        f.z += a.z * w00 + bq.z * w01 + c.z * w10 + d.z * w11;
        f.w += a.w * w00 + bq.w * w01 + c.w * w10 + d.w * w11;
      }
    }
    if (cnt == 0.f) continue;                                             // masked key (EU:252-258 attn_mask)
    const float inv = 1.f / (cnt + 1e-10f);                               // EU:249 masked multi-view mean
    f.x *= inv; f.y *= inv; f.z *= inv; f.w *= inv;
    float s = q4.x * f.x + q4.y * f.y + q4.z * f.z + q4.w * f.w;
#pragma unroll
    for (int o = LPG >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, LPG);
    const float m_new = fmaxf(m_run, s);
    const float alpha = expf(m_run - m_new), pw = expf(s - m_new);        // alpha = 0 on the first valid key
    l_run = l_run * alpha + pw;
    acc.x = acc.x * alpha + pw * f.x;
    acc.y = acc.y * alpha + pw * f.y;
    acc.z = acc.z * alpha + pw * f.z;
    acc.w = acc.w * alpha + pw * f.w;
    m_run = m_new;
    any = true;
  }
  if (active && lane_on) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (any) {
      const float inv = 1.f / l_run;
      o = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
    *reinterpret_cast<float4*>(p.ctx + ((long long)b * HW + cell) * p.Ci + coff) = o;
    if (sub == 0) p.valid[(long long)b * HW + cell] = any ? 1 : 0;
  }
}

}  // namespace

extern "C" int ff3d_cam_sample(const float* img_cl, const float* lidar2img, const float* img_aug, const float* qk,
                               float* ctx, uint8_t* valid, int B, int Ncam, int Ci, int Hi, int Wi, int H, int W,
                               int Z, const float* range_host, const float* input_hw_host, ff3d_stream_t stream) {
  FF3D_REQUIRE(img_cl && lidar2img && qk && ctx && valid && range_host && input_hw_host, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && B <= 65535 && Ncam > 0 && Ncam <= 8 && Ci > 0 && Hi > 0 && Wi > 0 && H > 0 && W > 0 &&
                   Z > 0 && Z <= 32,
               FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(Ci % 4 == 0 && Ci <= 256, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(img_cl) && ff3d_aligned16(qk) && ff3d_aligned16(ctx), FF3D_ERR_ALIGNMENT);
  CamParams p;
  p.img_cl = img_cl; p.lidar2img = lidar2img; p.img_aug = img_aug; p.qk = qk; p.ctx = ctx; p.valid = valid;
  p.Ncam = Ncam; p.Ci = Ci; p.Hi = Hi; p.Wi = Wi; p.H = H; p.W = W; p.Z = Z;
  for (int i = 0; i < 3; ++i) {
    p.lo[i] = range_host[i];
    p.hi[i] = range_host[3 + i];
  }
  p.in_h = input_hw_host[0];
  p.in_w = input_hw_host[1];
  const int C4 = Ci / 4;
  int lpg = 1;
  while (lpg < C4) lpg <<= 1;
  const int ppb = 256 / lpg;
  const dim3 grid((H * W + ppb - 1) / ppb, B), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  ff3d_clear_error();
  switch (lpg) {
    case 1: hipLaunchKernelGGL(cam_sample_kernel<1>, grid, block, 0, s, p, C4); break;
    case 2: hipLaunchKernelGGL(cam_sample_kernel<2>, grid, block, 0, s, p, C4); break;
    case 4: hipLaunchKernelGGL(cam_sample_kernel<4>, grid, block, 0, s, p, C4); break;
    case 8: hipLaunchKernelGGL(cam_sample_kernel<8>, grid, block, 0, s, p, C4); break;
    case 16: hipLaunchKernelGGL(cam_sample_kernel<16>, grid, block, 0, s, p, C4); break;
    case 32: hipLaunchKernelGGL(cam_sample_kernel<32>, grid, block, 0, s, p, C4); break;
    case 64: hipLaunchKernelGGL(cam_sample_kernel<64>, grid, block, 0, s, p, C4); break;
    default: return FF3D_ERR_BAD_SHAPE;
  }
  return ff3d_launch_status();
}
