"""Halo conv micro-benchmark (B=32, 256 -> 256, 180 x 180, fp32 NCHW output and pair output) + error vs fp64 of a crop.
Run once per variant (the schedule is chosen at first launch from the environment): FF3D_HALO_PP=0|1."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops


def t(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


B, C = int(os.environ.get('B', 32)), 256
H, W = int(os.environ.get('H', 180)), int(os.environ.get('W', 180))
g = torch.Generator().manual_seed(0)
x = torch.randn(B, C, H, W, generator=g).cuda()
w = (torch.randn(C, C, 3, 3, generator=g) * 0.02).cuda()
b = torch.randn(C, generator=g).cuda()
xp = ops.split_f16(x, to_nhwc=True)
wp = ops.split_weight_f16(w, bias=b)
out = ops.conv3x3_f16x3(xp, wp, b, relu=True)
ref = F.relu(F.conv2d(x[:1, :, :40, :70].double(), w.double(), b.double(), padding=1))[:, :, :39, :69]
err = float((out[:1, :, :39, :69].double() - ref).abs().max() / ref.abs().max())
pair = ops.conv3x3_f16x3(xp, wp, b, relu=True, split_out=True)
perr = None
if pair is not None:
    pv = ops.unsplit_f16(pair.map(lambda t_: t_.reshape(B * H * W, -1)), B, H, W)
    if pv is not None:
        perr = float((pv[:1, :, :39, :69].double() - ref).abs().max() / ref.abs().max())
ms = t(lambda: ops.conv3x3_f16x3(xp, wp, b, relu=True))
msp = t(lambda: ops.conv3x3_f16x3(xp, wp, b, relu=True, split_out=True))
fl = 2 * B * H * W * C * C * 9 * 3
print('GEO=%s TAP2=%s %dx%d ' % (os.environ.get('FF3D_HALO_GEO', 'auto'), os.environ.get('FF3D_HALO_TAP2', '0'), H, W), end='')
print('PP=%s B=%d  nchw %.3f ms (%.0f TF fp16-pass)  pair %s ms  err %.2e  pair err %s' % (
    os.environ.get('FF3D_HALO_PP', 'default'), B, ms, fl / 1e9 / ms, None if msp is None else '%.3f' % msp, err,
    None if perr is None else '%.2e' % perr))
