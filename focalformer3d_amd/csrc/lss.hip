// Fused lift-splat for gfx950 (Lift-Splat-Shoot camera branch, projects/mmdet3d_plugin/models/necks/lss.py).
//
// The reference materialises the outer product depth[pixel, d] * feat[pixel, :] for every frustum point
// (B*N*D*fH*fW x C floats: 1.4 GB per frame at 6 x 41 x 112 x 200 x 64, lss.py:135-141), filters, sorts and gathers
// that tensor, and only then sums it per BEV cell (lss.py:324-362 / the bev_pool extension).  Here the outer product is
// never formed: the frustum points are sorted by BEV cell once (indices only), and each cell's interval is reduced
// straight from the per-pixel feature rows (C floats, L2-resident: 34 MB per frame) and the depth probabilities:
//     out[cell, :] = sum_{(pixel, d) in cell} depth[pixel, d] * feat[pixel, :]
// Two kernels.  lss_cells_kernel fuses the whole frustum geometry (lss.py:232-276: undo image augmentation, un-project
// with the depth, camera -> ego, optional extra transform) and the voxel binning (lss.py:324-335) into one pass that
// emits a 4-byte sort key (cell id, or n_cells for a point outside the grid) per frustum point - the reference builds
// seven (B,N,D,H,W,3[,3]) temporaries with batched 3x3 matmuls for this.  The caller sorts the keys (stable radix sort) and
// derives the per-cell offsets table by binary search, all at static shapes (no host synchronisation).
// lss_splat_kernel: one wave per cell; C/4 lanes cover a feature row with 16-byte loads and the 64/(C/4) lane groups
// take different points of the cell in parallel, combined at the end with xor-shuffles; empty cells are written as
// zeros, so the output needs no memset.  HBM/L2-bound gather, no MFMA.
#include "ff3d_common.h"

namespace {

struct LssCellParams {
  const float *rots, *trans, *post_inv, *post_trans, *extra_rots, *extra_trans, *xs, *ys, *ds;
  int* keys;
  int B, N, D, fH, fW;
  float lo[3], dx[3];
  int nx[3];
};

__device__ __forceinline__ void mat3_vec(const float* __restrict__ m, float& x, float& y, float& z) {
  const float a = m[0] * x + m[1] * y + m[2] * z;
  const float b = m[3] * x + m[4] * y + m[5] * z;
  const float c = m[6] * x + m[7] * y + m[8] * z;
  x = a, y = b, z = c;
}

__global__ __launch_bounds__(256) void lss_cells_kernel(LssCellParams p) {
  const long long total = (long long)p.B * p.N * p.fH * p.fW * p.D;
  const int n_cells = p.B * p.nx[2] * p.nx[0] * p.nx[1];
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int d = (int)(e % p.D);
    long long pix = e / p.D;
    const int w = (int)(pix % p.fW);
    pix /= p.fW;
    const int h = (int)(pix % p.fH);
    const int cam = (int)(pix / p.fH), b = cam / p.N;
    float x = p.xs[w], y = p.ys[h], z = p.ds[d];
    if (p.post_trans) x -= p.post_trans[cam * 3], y -= p.post_trans[cam * 3 + 1], z -= p.post_trans[cam * 3 + 2];
    if (p.post_inv) mat3_vec(p.post_inv + cam * 9, x, y, z);
    x *= z, y *= z;                                               // pixel * depth (lss.py:251-254)
    mat3_vec(p.rots + cam * 9, x, y, z);
    x += p.trans[cam * 3], y += p.trans[cam * 3 + 1], z += p.trans[cam * 3 + 2];
    if (p.extra_rots) mat3_vec(p.extra_rots + cam * 9, x, y, z);
    if (p.extra_trans) x += p.extra_trans[cam * 3], y += p.extra_trans[cam * 3 + 1], z += p.extra_trans[cam * 3 + 2];
    const float fx = (x - p.lo[0]) / p.dx[0], fy = (y - p.lo[1]) / p.dx[1], fz = (z - p.lo[2]) / p.dx[2];
    int key = n_cells;
    // .long() truncates toward zero, so (-1, 0) lands in cell 0 exactly as in the reference (lss.py:327, 334-337)
    if (fabsf(fx) < 1e9f && fabsf(fy) < 1e9f && fabsf(fz) < 1e9f) {
      const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
      if (cx >= 0 && cx < p.nx[0] && cy >= 0 && cy < p.nx[1] && cz >= 0 && cz < p.nx[2])
        key = ((b * p.nx[2] + cz) * p.nx[0] + cx) * p.nx[1] + cy;
    }
    p.keys[e] = key;
  }
}

template <int LPG>
__global__ __launch_bounds__(256) void lss_splat_kernel(const float* __restrict__ feat, long long feat_ld,
                                                        const float* __restrict__ depth, int D,
                                                        const int* __restrict__ src, const int* __restrict__ offsets,
                                                        float* __restrict__ out, int C, int n_cells) {
  constexpr int G = 64 / LPG;
  const int lane = threadIdx.x & 63, sub = lane % LPG, grp = lane / LPG;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
  const bool on = sub * 4 < C;
  for (int cell = wave; cell < n_cells; cell += nwaves) {
    const int s = offsets[cell], len = offsets[cell + 1] - s;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = grp; i < len; i += G) {
      const int e = src[s + i];
      const float pr = depth[e];
      if (on) {
        const float4 r = *reinterpret_cast<const float4*>(feat + (long long)(e / D) * feat_ld + sub * 4);
        acc.x = fmaf(pr, r.x, acc.x);
        acc.y = fmaf(pr, r.y, acc.y);
        acc.z = fmaf(pr, r.z, acc.z);
        acc.w = fmaf(pr, r.w, acc.w);
      }
    }
    if (len > 1) {                                                // wave-uniform
#pragma unroll
      for (int o = LPG; o < 64; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o);
        acc.y += __shfl_xor(acc.y, o);
        acc.z += __shfl_xor(acc.z, o);
        acc.w += __shfl_xor(acc.w, o);
      }
    }
    if (grp == 0 && on) *reinterpret_cast<float4*>(out + (long long)cell * C + sub * 4) = acc;
  }
}

}  // namespace

extern "C" int ff3d_lss_cells(const float* rots, const float* trans, const float* post_rots_inv, const float* post_trans,
                              const float* extra_rots, const float* extra_trans, const float* xs, const float* ys,
                              const float* ds, int B, int N, int D, int fH, int fW, const float* lower_host,
                              const float* dx_host, const int32_t* nx_host, int32_t* keys, ff3d_stream_t stream) {
  FF3D_REQUIRE(rots && trans && xs && ys && ds && lower_host && dx_host && nx_host && keys, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && N > 0 && D > 0 && fH > 0 && fW > 0, FF3D_ERR_BAD_SHAPE);
  LssCellParams p{rots, trans, post_rots_inv, post_trans, extra_rots, extra_trans, xs, ys, ds, keys, B, N, D, fH, fW};
  long long cells = B;
  for (int i = 0; i < 3; ++i) {
    FF3D_REQUIRE(nx_host[i] > 0 && dx_host[i] > 0.f, FF3D_ERR_BAD_SHAPE);
    p.lo[i] = lower_host[i], p.dx[i] = dx_host[i], p.nx[i] = nx_host[i];
    cells *= nx_host[i];
  }
  const long long total = (long long)B * N * fH * fW * D;
  FF3D_REQUIRE(cells < (1ll << 31) - 1 && total < (1ll << 31), FF3D_ERR_BAD_SHAPE);   // keys / entry ids are int32
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  ff3d_clear_error();
  hipLaunchKernelGGL(lss_cells_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return ff3d_launch_status();
}

extern "C" int ff3d_lss_splat(const float* feat, int64_t feat_ld, const float* depth, int D, const int32_t* src,
                              const int32_t* cell_offsets, float* out, int C, int n_cells, ff3d_stream_t stream) {
  FF3D_REQUIRE(feat && depth && src && cell_offsets && out, FF3D_ERR_NULL);
  FF3D_REQUIRE(D > 0 && C > 0 && C % 4 == 0 && C <= 256 && n_cells > 0 && feat_ld >= C && feat_ld % 4 == 0,
               FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(ff3d_aligned16(feat) && ff3d_aligned16(out), FF3D_ERR_ALIGNMENT);
  int lpg = 1;
  while (lpg * 4 < C) lpg <<= 1;
  int blocks = (n_cells + 3) / 4;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipStream_t s = static_cast<hipStream_t>(stream);
  ff3d_clear_error();
#define FF3D_LSS_CASE(N)                                                                                          \
  case N:                                                                                                         \
    hipLaunchKernelGGL(lss_splat_kernel<N>, dim3(blocks), dim3(256), 0, s, feat, (long long)feat_ld, depth, D, src, \
                       cell_offsets, out, C, n_cells);                                                            \
    break;
  switch (lpg) {
    FF3D_LSS_CASE(1)
    FF3D_LSS_CASE(2)
    FF3D_LSS_CASE(4)
    FF3D_LSS_CASE(8)
    FF3D_LSS_CASE(16)
    FF3D_LSS_CASE(32)
    FF3D_LSS_CASE(64)
    default:
      return FF3D_ERR_BAD_SHAPE;
  }
#undef FF3D_LSS_CASE
  return ff3d_launch_status();
}
