"""Local attention micro-benchmark: scalar fp32 kernel vs the matrix-core pair kernel at 8 x C x 180 x 180, C = 32 .. 256."""
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


B, H, W = int(os.environ.get('B', 8)), 180, 180
for C in (32, 64, 128, 256):
    g = torch.Generator(device='cuda').manual_seed(0)
    q, k, v = (torch.randn(B, C, H, W, device='cuda', generator=g) for _ in range(3))
    rows = lambda x: ops.split_f16(x, to_nhwc=True).map(lambda p_: p_.reshape(B * H * W, C))
    qp, kp, vp = rows(q), rows(k), rows(v)
    a = ops.local_attention(q, k, v, 9, C ** -0.5)
    b = ops.local_attention_pair(qp, kp, vp, B, H, W, 9, C ** -0.5).value().view(B, H, W, C).permute(0, 3, 1, 2)
    print('C=%3d  scalar %.3f ms   MFMA pair (pre-pass + attention) %.3f ms   max diff %.2e' % (
        C, t(lambda: ops.local_attention(q, k, v, 9, C ** -0.5)), t(lambda: ops.local_attention_pair(qp, kp, vp, B, H, W, 9, C ** -0.5)),
        float((a - b).abs().max())))
