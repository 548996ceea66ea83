"""RoI sampler micro-benchmark at the head's shape (B x 600 boxes, 3 levels, C=256, pair output)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops
def t(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B, C, Nq = int(os.environ.get('B', 32)), 256, 600
g = torch.Generator(device='cuda').manual_seed(0)
hw = [(180, 180), (90, 90), (45, 45)]
raw = torch.randn(B, sum(h * w for h, w in hw), C, device='cuda', generator=g)
box = torch.randn(B, 10, Nq, device='cuda', generator=g)
box[:, 0:2] = torch.rand(B, 2, Nq, device='cuda', generator=g) * 180
box[:, 3:6] = box[:, 3:6] * 0.3 + torch.tensor([0.6, 1.5, 0.5], device='cuda')[None, :, None]      # log sizes: ~1.8 x 4.5 m
if os.environ.get('SMALL') == '1':                 # ~1 m boxes: what a randomly initialised head regresses (the bench's workload)
    box[:, 3:6] = torch.randn(B, 3, Nq, device='cuda', generator=g) * 0.05
coder = (8, 0.075, 0.075, -54.0, -54.0)
rng = (-54.0, -54.0, 54.0, 54.0)
f = lambda dt: ops.roi_grid_sample(raw, hw, box, 7, 1.2, coder, rng, layout=1, out_dtype=dt)
a = f(torch.float32)
print('LDS=%s SMALL=%s ' % (os.environ.get('FF3D_ROI_LDS', 'default'), os.environ.get('SMALL', '0')), end='')
print('B=%d roi sampler: pair out %.3f ms, fp32 out %.3f ms, checksum %.6e' % (B, t(lambda: f('f16split')), t(lambda: f(torch.float32)), float(a.double().sum())))
