"""MSDA backward micro-benchmark at the training step's shape (B=4, 720 queries, 8 heads x 32, 3 levels x 4 points)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops
def t(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B, Nq, H, D = int(os.environ.get('B', 4)), 720, 8, 32
hw = [(180, 180), (90, 90), (45, 45)]
Nv = sum(h * w for h, w in hw)
g = torch.Generator(device='cuda').manual_seed(0)
value = torch.randn(B, Nv, H, D, device='cuda', generator=g)
loc = torch.rand(B, Nq, H, 3, 4, 2, device='cuda', generator=g)
w = torch.softmax(torch.randn(B, Nq, H, 12, device='cuda', generator=g), -1).view(B, Nq, H, 3, 4)
go = torch.randn(B, Nq, H * D, device='cuda', generator=g)
gv, gl, gw = ops.msda_bwd(value, hw, loc, w, go)
print('VN=%s B=%d: %.4f ms  checksums %.6e %.6e %.6e' % (os.environ.get('FF3D_MSDA_BWD_VN', '4'), B, t(lambda: ops.msda_bwd(value, hw, loc, w, go)),
      float(gv.double().sum()), float(gl.double().sum()), float(gw.double().sum())))
