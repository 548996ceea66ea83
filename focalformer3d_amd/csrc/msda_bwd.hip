// Multi-scale deformable attention BACKWARD for gfx950 - counterpart of mmcv 1.3.18 `ms_deform_attn_backward`
// (`ms_deformable_col2im_*` kernels of mmcv/ops/csrc/.../ms_deform_attn_cuda_kernel.cuh; un-vendored third party, restated
// from the published algorithm):
//     out[b,q,h,:] = sum_{l,p} w[b,q,h,l,p] * bilinear(value_l[b,:,h,:], loc[b,q,h,l,p])
//   grad_value[b,cell,h,:] += w * c_corner * grad_out[b,q,h,:]                        (atomic scatter)
//   grad_w[b,q,h,l,p]       = < grad_out[b,q,h,:], bilinear(...) >
//   grad_loc[b,q,h,l,p]     = w * (W_l * < grad_out, d bilinear / dx >,  H_l * < grad_out, d bilinear / dy >)
// Same work decomposition as the forward kernel (msda.hip): one (batch, query, head) pair per LPG adjacent lanes, each
// lane owning 4 (or 1) channels; the per-pair locations / weights are staged through LDS; the three inner products per
// sampling point are reduced over the pair's lanes with xor-shuffles.  The value gradient is scattered with fp32 global
// atomics (order-dependent rounding, as in the reference kernel).
#include <cstdlib>

#include "ff3d_common.h"

namespace {

struct MsdaBwdParams {
  const float *value, *loc, *attn_w, *grad_out;
  float *grad_value, *grad_loc, *grad_w;
  int npairs, Nq, heads, Dh, P, LP;
  LevelTable lv;
};

template <int LPG, int VN>
__global__ __launch_bounds__(256) void msda_bwd_kernel(MsdaBwdParams p) {
  constexpr int PPB = 256 / LPG;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_loc = smem;                  // [PPB][LP][2]
  float* s_w = smem + PPB * p.LP * 2;   // [PPB][LP]
  const unsigned bid = ff3d_xcd_remap(blockIdx.x, gridDim.x);
  const int pair0 = bid * PPB, npair_blk = min(PPB, p.npairs - pair0), LP = p.LP, tid = threadIdx.x;
  for (int i = tid; i < npair_blk * LP * 2; i += 256) s_loc[i] = p.loc[(long long)pair0 * LP * 2 + i];
  for (int i = tid; i < npair_blk * LP; i += 256) s_w[i] = p.attn_w[(long long)pair0 * LP + i];
  __syncthreads();

  const int pl = tid / LPG, sub = tid - pl * LPG;
  const bool live = pl < npair_blk;                       // dead lanes still take part in the shuffles
  const int pair = pair0 + (live ? pl : 0);
  const int row = pair / p.heads, h = pair - row * p.heads, b = row / p.Nq;
  const float* ploc = s_loc + (live ? pl : 0) * LP * 2;
  const float* pw = s_w + (live ? pl : 0) * LP;
  const long long stride = (long long)p.heads * p.Dh;
  const long long base = (long long)b * p.lv.Nv * stride + h * p.Dh + sub * VN;
  float g[VN];
#pragma unroll
  for (int c = 0; c < VN; ++c) g[c] = live ? p.grad_out[(long long)pair * p.Dh + sub * VN + c] : 0.f;

  for (int l = 0; l < p.lv.L; ++l) {
    const int Hl = p.lv.H[l], Wl = p.lv.W[l];
    const long long lbase = base + (long long)p.lv.start[l] * stride;
    const float fH = (float)Hl, fW = (float)Wl;
    for (int pt = 0; pt < p.P; ++pt) {
      const int k = l * p.P + pt;
      const float x = ploc[2 * k], y = ploc[2 * k + 1], aw = pw[k];
      const float w_im = fminf(fmaxf(x * fW - 0.5f, -2.f), fW + 1.f);
      const float h_im = fminf(fmaxf(y * fH - 0.5f, -2.f), fH + 1.f);
      const float h_lo = floorf(h_im), w_lo = floorf(w_im);
      const float lh = h_im - h_lo, lw = w_im - w_lo, hh = 1.f - lh, hw = 1.f - lw;
      const int y0 = (int)h_lo, x0 = (int)w_lo, y1 = y0 + 1, x1 = x0 + 1;
      const bool vy0 = (unsigned)y0 < (unsigned)Hl, vy1 = (unsigned)y1 < (unsigned)Hl;
      const bool vx0 = (unsigned)x0 < (unsigned)Wl, vx1 = (unsigned)x1 < (unsigned)Wl;
      const bool ok[4] = {live && vy0 && vx0, live && vy0 && vx1, live && vy1 && vx0, live && vy1 && vx1};
      const int cy[2] = {min(max(y0, 0), Hl - 1), min(max(y1, 0), Hl - 1)};
      const int cx[2] = {min(max(x0, 0), Wl - 1), min(max(x1, 0), Wl - 1)};
      const float cw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
      float d[4];                                          // <grad_out slice, corner value slice>
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const long long o = lbase + (long long)(cy[c4 >> 1] * Wl + cx[c4 & 1]) * stride;
        float acc = 0.f;
        if (ok[c4]) {
#pragma unroll
          for (int c = 0; c < VN; ++c) {
            acc = fmaf(g[c], p.value[o + c], acc);
            atomicAdd(p.grad_value + o + c, cw[c4] * aw * g[c]);
          }
        }
        d[c4] = acc;
      }
      float gval = cw[0] * d[0] + cw[1] * d[1] + cw[2] * d[2] + cw[3] * d[3];
      float gx = hh * (d[1] - d[0]) + lh * (d[3] - d[2]);  // d bilinear / d w_im
      float gy = hw * (d[2] - d[0]) + lw * (d[3] - d[1]);  // d bilinear / d h_im
#pragma unroll
      for (int o = LPG >> 1; o > 0; o >>= 1) {
        gval += __shfl_xor(gval, o);
        gx += __shfl_xor(gx, o);
        gy += __shfl_xor(gy, o);
      }
      if (live && sub == 0) {
        const long long e = (long long)pair * LP + k;
        p.grad_w[e] = gval;
        p.grad_loc[2 * e] = gx * aw * fW;
        p.grad_loc[2 * e + 1] = gy * aw * fH;
      }
    }
  }
}

template <int VN>
int launch_bwd(int lpg, const MsdaBwdParams& p, hipStream_t s) {
  const int ppb = 256 / lpg;
  const size_t smem = (size_t)ppb * p.LP * 3 * sizeof(float);
  if (smem > 64 * 1024) return FF3D_ERR_UNSUPPORTED;
  const unsigned grid = (p.npairs + ppb - 1) / ppb;
  ff3d_clear_error();
#define FF3D_BWD_CASE(N)                                                                \
  case N:                                                                               \
    hipLaunchKernelGGL((msda_bwd_kernel<N, VN>), dim3(grid), dim3(256), smem, s, p);    \
    break;
  switch (lpg) {
    FF3D_BWD_CASE(1)
    FF3D_BWD_CASE(2)
    FF3D_BWD_CASE(4)
    FF3D_BWD_CASE(8)
    FF3D_BWD_CASE(16)
    FF3D_BWD_CASE(32)
    FF3D_BWD_CASE(64)
    default:
      return FF3D_ERR_BAD_SHAPE;
  }
#undef FF3D_BWD_CASE
  return ff3d_launch_status();
}

}  // namespace

extern "C" int ff3d_msda_bwd(const float* value, const float* sampling_loc, const float* attn_w, const float* grad_out,
                             float* grad_value, float* grad_sampling_loc, float* grad_attn_w, int B, int Nv, int Nq,
                             int heads, int Dh, int L, int P, const int32_t* level_hw_host, ff3d_stream_t stream) {
  FF3D_REQUIRE(value && sampling_loc && attn_w && grad_out && grad_value && grad_sampling_loc && grad_attn_w, FF3D_ERR_NULL);
  FF3D_REQUIRE(B > 0 && Nv > 0 && Nq > 0 && heads > 0 && Dh > 0 && L > 0 && P > 0, FF3D_ERR_BAD_SHAPE);
  FF3D_REQUIRE(L <= FF3D_MAX_LEVELS && L * P <= 64 && (long long)B * Nq * heads < (1ll << 31), FF3D_ERR_BAD_SHAPE);
  // One channel per lane whenever Dh itself is a legal lane-group size: the value-gradient atomics of a corner are then Dh
  // consecutive floats (0.14 vs 0.47 ms at 4 frames x 720 queries x 8 heads x 32 with four channels per lane, whose atomics
  // interleave 16 bytes apart); tuning hook FF3D_MSDA_BWD_VN=4 restores the wide form.
  static const int vn_env = [] {
    const char* e = getenv("FF3D_MSDA_BWD_VN");
    return e ? atoi(e) : 0;
  }();
  const bool pow2 = Dh <= 64 && (Dh & (Dh - 1)) == 0;
  const int vn = (pow2 && vn_env != 4) ? 1 : (Dh % 4 == 0 ? 4 : 1), lpg = Dh / vn;
  FF3D_REQUIRE(lpg <= 64 && (lpg & (lpg - 1)) == 0, FF3D_ERR_BAD_SHAPE);
  MsdaBwdParams p;
  FF3D_REQUIRE(ff3d_make_levels(level_hw_host, L, &p.lv) && p.lv.Nv == Nv, FF3D_ERR_BAD_SHAPE);
  p.value = value; p.loc = sampling_loc; p.attn_w = attn_w; p.grad_out = grad_out;
  p.grad_value = grad_value; p.grad_loc = grad_sampling_loc; p.grad_w = grad_attn_w;
  p.npairs = B * Nq * heads; p.Nq = Nq; p.heads = heads; p.Dh = Dh; p.P = P; p.LP = L * P;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return vn == 4 ? launch_bwd<4>(lpg, p, s) : launch_bwd<1>(lpg, p, s);
}
