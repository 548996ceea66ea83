// RoI grid feature sampler for gfx950: box decode -> g x g grid in the box frame -> bilinear
// sampling of every pyramid level from the channels-last (B, Nv, C) pyramid.  One block per
// (frame, query); C/4 lanes cover the channels of one sampling point with 16-byte loads, so each
// bilinear corner is one contiguous C*4-byte row.  HBM-bound gather + (B*Nq, L*C*g*g) write.
// (Tried: each lane group walking a contiguous run of grid points and keeping the previous point's four corner rows in
// registers - neighbouring points mostly share cells, 11.5 GB of L2 corner reads per launch at batch 32 - 3.72 vs 1.30 ms: the
// reuse test serialises the loads of consecutive points that the strided walk keeps in flight together.
//  Round 6, also slower: the cells of a box staged once per (query, level) through LDS - bounding cell rectangle of the g x g points by
//  LDS min / max, its rows copied in with 16-byte loads when they fit 40 KB, corners read with ds_read_b128, direct loads otherwise.
//  It removes ~9/10 of the 11.5 GB of L2 corner reads, and costs 1.67 - 1.73 vs 1.07 - 1.30 ms (1 m boxes / car-sized boxes, 32 frames,
//  profiles/r06_h_roi_lds_ab.txt): six block-wide barriers and a global -> register -> LDS round trip per level serialise what eight
//  resident blocks of independent loads per CU overlap; the step 1258 - 1261 vs 1304 - 1311 frames/s.
//  Also level / slower: the corner loads of 2 - 4 items issued before the first is consumed (as msda.hip's PT == 4 path): pair output
//  1.14 - 1.19 vs 1.10 - 1.12 ms for 1 m boxes, 1.28 - 1.33 vs 1.28 - 1.31 for car-sized ones, step level (profiles/r06_t_roi_batched_loads_ab.txt):
//  at eight waves per SIMD the loads of different waves already overlap.)
//
// Backward (training path): roi_grid_sample_bwd_kernel scatters the gradient of the RoI matrix back into
// the channels-last pyramid with the same geometry; lanes run over consecutive channels, so every bilinear corner is one
// contiguous run of hardware fp32 atomic adds (the framework's grid_sample backward walks the channels of an NCHW map with a
// stride of H*W floats per atomic: 10.3 ms per call at B=4, C=256 on this GPU, 30 % of a training step).
#include "ff3d_common.h"

namespace {

struct RoiParams {
  const float* feat_cl;
  const float* query_box;
  void* out;
  float* grid_out;
  int out_bf16;          // 0 fp32, 1 bf16, 2 split fp16 (hi plane, then lo' plane)
  long long out_plane;   // halves per plane (split output)
  LevelTable lv;
  int Nq, C, g, box_dim, layout;
  float expand;
  float osf, vx, vy, pcx, pcy;         // bbox coder (BC:56-57)
  float lo_x, lo_y, hi_x, hi_y;        // FD:903-906
  const int* feat_exp;                 // split output: bound exponent of feat_cl = exponent of the RoI pair (ff3d.h)
  const float* grad_out;               // backward: gradient of the RoI matrix, same layout as `out`
  float* grad_feat;                    // backward: (B, Nv, C), zero-initialised by the caller
};

// Normalised sampling point `t` (< g*g) of the box of one query, FD:891-909.
__device__ __forceinline__ void roi_point(const RoiParams& p, const float* box, int t, float& nx, float& ny) {
  const float cx = box[0] * p.osf * p.vx + p.pcx;                // BC:56-57
  const float cy = box[(long long)p.Nq] * p.osf * p.vy + p.pcy;
  const float w = expf(box[3ll * p.Nq] * p.expand);              // FD:892: dim*expand before exp (BC:59-61)
  const float l = expf(box[4ll * p.Nq] * p.expand);
  const float yaw = atan2f(box[6ll * p.Nq], box[7ll * p.Nq]);    // BC:64-65
  const int i = t / p.g, j = t - i * p.g;                        // FD:1657-1663, first index slow
  const float px = ((float)i + 0.5f) / (float)p.g * w - w / 2.f;
  const float py = ((float)j + 0.5f) / (float)p.g * l - l / 2.f;
  const float c = cosf(yaw), s = sinf(yaw);
  const float rx = px * c + py * s + cx;                         // mmdet3d 0.17.1 rotation_3d_in_axis, axis 2
  const float ry = -px * s + py * c + cy;
  nx = (rx - p.lo_x) / (p.hi_x - p.lo_x) * 2.f - 1.f;            // FD:907-909
  ny = (ry - p.lo_y) / (p.hi_y - p.lo_y) * 2.f - 1.f;
  nx = fminf(fmaxf(nx, -2.f), 2.f);
  ny = fminf(fmaxf(ny, -2.f), 2.f);
}

// F.grid_sample(bilinear, zeros, align_corners=False) of one point on an (Hl, Wl) level: the four (clamped) cells and their
// weights (0 outside the map).
struct Bilinear {
  int cell[4];
  float w[4];
};
__device__ __forceinline__ Bilinear roi_bilinear(float gx, float gy, int Hl, int Wl) {
  // unnormalise, align_corners=False: ((coord + 1) * size - 1) / 2
  const float ix = fminf(fmaxf(((gx + 1.f) * (float)Wl - 1.f) / 2.f, -2.f), (float)Wl + 1.f);
  const float iy = fminf(fmaxf(((gy + 1.f) * (float)Hl - 1.f) / 2.f, -2.f), (float)Hl + 1.f);
  const float x0f = floorf(ix), y0f = floorf(iy);
  const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  const float lx = ix - x0f, ly = iy - y0f, hx = 1.f - lx, hy = 1.f - ly;
  const bool vx0 = (unsigned)x0 < (unsigned)Wl, vx1 = (unsigned)x1 < (unsigned)Wl;
  const bool vy0 = (unsigned)y0 < (unsigned)Hl, vy1 = (unsigned)y1 < (unsigned)Hl;
  Bilinear r;
  r.w[0] = (vy0 && vx0) ? hy * hx : 0.f;
  r.w[1] = (vy0 && vx1) ? hy * lx : 0.f;
  r.w[2] = (vy1 && vx0) ? ly * hx : 0.f;
  r.w[3] = (vy1 && vx1) ? ly * lx : 0.f;
  const int cx0 = min(max(x0, 0), Wl - 1), cx1 = min(max(x1, 0), Wl - 1);
  const int cy0 = min(max(y0, 0), Hl - 1), cy1 = min(max(y1, 0), Hl - 1);
  r.cell[0] = cy0 * Wl + cx0;
  r.cell[1] = cy0 * Wl + cx1;
  r.cell[2] = cy1 * Wl + cx0;
  r.cell[3] = cy1 * Wl + cx1;
  return r;
}

__global__ __launch_bounds__(256) void roi_grid_sample_kernel(RoiParams p) {
  __shared__ float s_gx[256], s_gy[256];
  const int row = blockIdx.x;  // b*Nq + q
  const int b = row / p.Nq, q = row - b * p.Nq;
  const int G = p.g * p.g, tid = threadIdx.x;
  const float* box = p.query_box + (long long)b * p.box_dim * p.Nq + q;

  // ---- sampling grid (every thread computes the box; threads < G compute one grid point each)
  if (tid < G) {
    float nx, ny;
    roi_point(p, box, tid, nx, ny);
    s_gx[tid] = nx;
    s_gy[tid] = ny;
    if (p.grid_out) {
      p.grid_out[((long long)row * G + tid) * 2] = nx;
      p.grid_out[((long long)row * G + tid) * 2 + 1] = ny;
    }
  }
  __syncthreads();

  const int C4 = p.C >> 2;                // float4 lanes per point
  const int pts_par = 256 / C4 > 0 ? 256 / C4 : 1;
  const int lane_c = tid % C4, lane_p = tid / C4;
  const long long out_row = (long long)row * p.lv.L * p.C * G;
  const int items = p.lv.L * G;
  for (int it = lane_p; it < items; it += pts_par) {
    if (tid >= pts_par * C4) break;
    const int l = it / G, gi = it - l * G;
    const int Hl = p.lv.H[l], Wl = p.lv.W[l];
    const Bilinear bl = roi_bilinear(s_gx[gi], s_gy[gi], Hl, Wl);
    const float w00 = bl.w[0], w01 = bl.w[1], w10 = bl.w[2], w11 = bl.w[3];
    const float* base = p.feat_cl + ((long long)b * p.lv.Nv + p.lv.start[l]) * p.C + lane_c * 4;
    const float4 a = *reinterpret_cast<const float4*>(base + (long long)bl.cell[0] * p.C);
    const float4 bb = *reinterpret_cast<const float4*>(base + (long long)bl.cell[1] * p.C);
    const float4 c = *reinterpret_cast<const float4*>(base + (long long)bl.cell[2] * p.C);
    const float4 d = *reinterpret_cast<const float4*>(base + (long long)bl.cell[3] * p.C);
    float4 r;
    r.x = a.x * w00 + bb.x * w01 + c.x * w10 + d.x * w11;
    r.y = a.y * w00 + bb.y * w01 + c.y * w10 + d.y * w11;
    r.z = a.z * w00 + bb.z * w01 + c.z * w10 + d.z * w11;
    r.w = a.w * w00 + bb.w * w01 + c.w * w10 + d.w * w11;
    if (p.out_bf16 == 2) {   // (hi, lo') fp16 pair for the split-fp16 GEMM (splitmm.hip), layout 1 only
      const float sc = ff3d_pow2(-ff3d_ld_exp(p.feat_exp));     // a bilinear sample is a convex combination of cells
      const float f[4] = {r.x * sc, r.y * sc, r.z * sc, r.w * sc};
      _Float16 hi[4], lo[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hi[k] = (_Float16)f[k];
        lo[k] = (_Float16)((f[k] - (float)hi[k]) * 2048.f);
      }
      _Float16* oh = reinterpret_cast<_Float16*>(p.out) + out_row + ((long long)l * G + gi) * p.C + lane_c * 4;
      *reinterpret_cast<uint2*>(oh) = *reinterpret_cast<uint2*>(hi);
      *reinterpret_cast<uint2*>(oh + p.out_plane) = *reinterpret_cast<uint2*>(lo);
    } else if (p.out_bf16) {     // bf16 output (round-to-nearest-even), layout 1 only: 8-byte stores
      auto rne = [](float f) -> unsigned {
        const unsigned u = __float_as_uint(f);
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
      };
      uint2 pk;
      pk.x = rne(r.x) | (rne(r.y) << 16);
      pk.y = rne(r.z) | (rne(r.w) << 16);
      unsigned short* ob = reinterpret_cast<unsigned short*>(p.out);
      *reinterpret_cast<uint2*>(ob + out_row + ((long long)l * G + gi) * p.C + lane_c * 4) = pk;
    } else if (p.layout == 1) {  // [level][point][channel]: coalesced 16-byte stores
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + out_row + ((long long)l * G + gi) * p.C + lane_c * 4) = r;
    } else {              // reference order [level][channel][point] (FD:919)
      float* o = reinterpret_cast<float*>(p.out) + out_row + ((long long)l * p.C + lane_c * 4) * G + gi;
      o[0] = r.x;
      o[G] = r.y;
      o[2 * G] = r.z;
      o[3 * G] = r.w;
    }
  }
}

// One block per (frame, query).  256 threads = (points in flight) x (consecutive channels): thread t adds the channels
// t % cpb, + cpb, ... of its point, so one wave's atomics of one corner are 64 consecutive floats of one pyramid row.
__global__ __launch_bounds__(256) void roi_grid_sample_bwd_kernel(RoiParams p) {
  __shared__ float s_gx[256], s_gy[256];
  const int row = blockIdx.x;
  const int b = row / p.Nq, q = row - b * p.Nq;
  const int G = p.g * p.g, tid = threadIdx.x;
  if (tid < G) roi_point(p, p.query_box + (long long)b * p.box_dim * p.Nq + q, tid, s_gx[tid], s_gy[tid]);
  __syncthreads();
  const int cpb = p.C < 256 ? p.C : 256;           // channels covered by one pass of a point's threads
  const int pts_par = 256 / cpb;
  const int lane_c = tid % cpb, lane_p = tid / cpb;
  if (lane_p >= pts_par) return;                   // C not a divisor of 256: the tail threads idle
  const long long g_row = (long long)row * p.lv.L * p.C * G;
  const int items = p.lv.L * G;
  for (int it = lane_p; it < items; it += pts_par) {
    const int l = it / G, gi = it - l * G;
    const Bilinear bl = roi_bilinear(s_gx[gi], s_gy[gi], p.lv.H[l], p.lv.W[l]);
    float* base = p.grad_feat + ((long long)b * p.lv.Nv + p.lv.start[l]) * p.C;
    for (int c = lane_c; c < p.C; c += cpb) {
      const float gv = p.layout == 1 ? p.grad_out[g_row + ((long long)l * G + gi) * p.C + c]
                                     : p.grad_out[g_row + ((long long)l * p.C + c) * G + gi];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (bl.w[k] != 0.f) atomicAdd(base + (long long)bl.cell[k] * p.C + c, bl.w[k] * gv);
    }
  }
}

}  // namespace

extern "C" int ff3d_roi_grid_sample(const float* feat_cl, const float* query_box, void* out, int out_dtype,
                                    float* grid_out, int B, int Nq, int C, int L, const int32_t* level_hw_host, int g,
                                    int box_dim, float expand, const float* coder_host, const float* range_host,
                                    int layout, const int32_t* feat_exp, ff3d_stream_t stream) {
  FF3D_REQUIRE(out_dtype == FF3D_F32 || ((out_dtype == FF3D_BF16 || out_dtype == FF3D_F16_SPLIT) && layout == 1),
               FF3D_ERR_BAD_DTYPE);
  FF3D_REQUIRE(feat_cl && query_box && out && coder_host && range_host, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && Nq > 0 && C > 0 && g > 0 && g * g <= 256 && box_dim >= 8, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(C % 4 == 0 && C <= 1024, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(layout == 0 || layout == 1, FF3D_ERR_UNSUPPORTED);
  FF3D_REQUIRE(ff3d_aligned16(feat_cl) && ff3d_aligned16(out), FF3D_ERR_ALIGNMENT);
  FF3D_REQUIRE((long long)B * Nq < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  RoiParams p = {};
  FF3D_REQUIRE(ff3d_make_levels(level_hw_host, L, &p.lv), FF3D_ERR_BAD_SHAPE);
  p.feat_cl = feat_cl;
  p.query_box = query_box;
  p.out = out;
  p.grid_out = grid_out;
  p.Nq = Nq;
  p.C = C;
  p.g = g;
  p.box_dim = box_dim;
  p.layout = layout;
  p.out_bf16 = out_dtype == FF3D_BF16 ? 1 : out_dtype == FF3D_F16_SPLIT ? 2 : 0;
  p.feat_exp = feat_exp;
  p.out_plane = ((long long)B * Nq + 1) * L * C * g * g;   // + the zero row of the split-GEMM operand contract
  p.expand = expand;
  p.osf = coder_host[0];
  p.vx = coder_host[1];
  p.vy = coder_host[2];
  p.pcx = coder_host[3];
  p.pcy = coder_host[4];
  p.lo_x = range_host[0];
  p.lo_y = range_host[1];
  p.hi_x = range_host[2];
  p.hi_y = range_host[3];
  ff3d_clear_error();
  hipLaunchKernelGGL(roi_grid_sample_kernel, dim3(B * Nq), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return ff3d_launch_status();
}

extern "C" int ff3d_roi_grid_sample_bwd(const float* grad_out, const float* query_box, float* grad_feat_cl, int B, int Nq,
                                        int C, int L, const int32_t* level_hw_host, int g, int box_dim, float expand,
                                        const float* coder_host, const float* range_host, int layout,
                                        ff3d_stream_t stream) {
  FF3D_REQUIRE(grad_out && query_box && grad_feat_cl && coder_host && range_host, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && Nq > 0 && C > 0 && g > 0 && g * g <= 256 && box_dim >= 8, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(layout == 0 || layout == 1, FF3D_ERR_UNSUPPORTED);
  FF3D_REQUIRE((long long)B * Nq < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  RoiParams p = {};
  FF3D_REQUIRE(ff3d_make_levels(level_hw_host, L, &p.lv), FF3D_ERR_BAD_SHAPE);
  p.grad_out = grad_out;
  p.query_box = query_box;
  p.grad_feat = grad_feat_cl;
  p.Nq = Nq;
  p.C = C;
  p.g = g;
  p.box_dim = box_dim;
  p.layout = layout;
  p.expand = expand;
  p.osf = coder_host[0];
  p.vx = coder_host[1];
  p.vy = coder_host[2];
  p.pcx = coder_host[3];
  p.pcy = coder_host[4];
  p.lo_x = range_host[0];
  p.lo_y = range_host[1];
  p.hi_x = range_host[2];
  p.hi_y = range_host[3];
  ff3d_clear_error();
  hipLaunchKernelGGL(roi_grid_sample_bwd_kernel, dim3(B * Nq), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return ff3d_launch_status();
}
