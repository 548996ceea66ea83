"""Round 6 experiment: the halo conv reading the caller's NCHW fp32 map (conversion to (hi, lo') pairs folded into the halo staging,
ff3d_exp_conv3x3_halo_nchwsrc of the EXPERIMENTS library) against [conversion pass + halo conv over the pair].
    FF3D_BUILD_EXPERIMENTS=1 python -m focalformer3d_amd.build
    FF3D_LIB=focalformer3d_amd/lib/libff3d_hip_exp.so B=32 H=180 W=180 python tools/experiments/exp_halo_nchw.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import _lib, ops  # noqa: E402


def t(fn, n=10, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


B, Cc = int(os.environ.get('B', 32)), int(os.environ.get('C', 256))
H, W = int(os.environ.get('H', 180)), int(os.environ.get('W', 180))
g = torch.Generator().manual_seed(0)
x = torch.randn(B, Cc, H, W, generator=g).cuda()
w = (torch.randn(Cc, Cc, 3, 3, generator=g) * 0.02).cuda()
b = torch.randn(Cc, generator=g).cuda()
wp = ops.split_weight_f16(w, bias=b)
hint = ops.new_hint(x.device)
xp = ops.split_f16(x, to_nhwc=True, hint=hint)
xp = ops.split_f16(x, to_nhwc=True, hint=hint)            # steady state: the guess holds
lib = _lib.load()
fn = lib.ff3d_exp_conv3x3_halo_nchwsrc
vp, i_ = C.c_void_p, C.c_int
fn.restype = i_
fn.argtypes = [vp, vp, vp, vp, vp, i_, vp, vp, vp, i_, i_, i_, i_, i_, i_, C.POINTER(_lib.Scale), vp]
wt = ops._halo_tiled_weight(wp, Cc, Cc)


hint2 = ops.new_hint(x.device)                               # the conv's own exponent record (first call: guess 0 -> flagged -> redo)


def nchw(split_out, nvar=9):
    sc, out_exp = ops._scale(None, wp, want_out=True)
    buf = ops._split_planes(B * H * W, Cc, x.device) if split_out else None
    out = None if split_out else torch.empty(B, Cc, H, W, device=x.device)
    st = fn(x.data_ptr(), hint2.data_ptr(), wt[0].data_ptr(), wt[1].data_ptr(), b.data_ptr(), 1, 0 if out is None else out.data_ptr(),
            buf[0].data_ptr() if split_out else 0, buf[1].data_ptr() if split_out else 0, B, Cc, H, W, Cc, nvar, sc,
            torch.cuda.current_stream().cuda_stream)
    _lib.check(st, 'ff3d_exp_conv3x3_halo_nchwsrc')
    return ops.Pair(buf[0, :-1].view(B, H, W, Cc), buf[1, :-1].view(B, H, W, Cc), out_exp) if split_out else out


PAIR_ONLY = os.environ.get('PAIR_ONLY') == '1'         # the 8 x 32 geometry experiment has the pair output form only
ref = ops.conv3x3_f16x3(xp, wp, b, relu=True)
got = ref if PAIR_ONLY else nchw(False)
print('fp32 output identical to the pair-input kernel:', bool(torch.equal(ref, got)), 'max abs diff %.3e' % float((ref - got).abs().max()))
refp = ops.conv3x3_f16x3(xp, wp, b, relu=True, split_out=True)
gotp = nchw(True)
print('pair output identical:', bool(torch.equal(refp[0], gotp[0]) and torch.equal(refp[1], gotp[1])))
ms_split = t(lambda: ops.split_f16(x, to_nhwc=True, hint=hint))
ms_conv = t(lambda: ops.conv3x3_f16x3(xp, wp, b, relu=True, split_out=True))
ms_both = t(lambda: ops.conv3x3_f16x3(ops.split_f16(x, to_nhwc=True, hint=hint), wp, b, relu=True, split_out=True))
ms_nchw = t(lambda: nchw(True))
ms_prod = t(lambda: ops.conv3x3_f16x3_nchwsrc(x, hint2, wp, b, relu=True, split_out=True))
ms_conv32 = t(lambda: ops.conv3x3_f16x3(xp, wp, b, relu=True))
ms_nchw32 = float('nan') if PAIR_ONLY else t(lambda: nchw(False))
print('B=%d %dx%dx%d  pair out: split %.3f + conv %.3f = %.3f (back to back %.3f) ms | conv over NCHW fp32 %.3f ms (ops wrapper %.3f)   fp32 out: conv %.3f | over NCHW %.3f'
      % (B, Cc, H, W, ms_split, ms_conv, ms_split + ms_conv, ms_both, ms_nchw, ms_prod, ms_conv32, ms_nchw32))
names = {9: 'shipped: pixel-fastest slots, top of the step', 0: 'DMA slot order, behind the first MFMA pass', 1: 'DMA slot order, top of the step',
         8: 'pixel-fastest, behind the first MFMA pass', 73: '9 + non-temporal requests', 137: '9 + requests ahead of the weight DMAs', 40: 'pixel-fastest, two register sets in flight', 2: 'WRONG results: no requests in the loop',
         6: 'WRONG results: neither requests nor conversion'}
for nv in ((9, 9) if PAIR_ONLY else (9, 73, 137, 9, 73, 137, 40, 0, 1, 8, 2, 6)):
    same = bool(torch.equal(nchw(True, nv)[0], refp[0]) and torch.equal(nchw(True, nv)[1], refp[1]))
    print('variant %2d (%s): %.3f ms (conv + check + guarded second launch), identical: %s' % (nv, names[nv], t(lambda: nchw(True, nv)), same))
