"""Self-attention micro-benchmark at the head's shape (B frames x 600 queries, 8 heads x 32) + error vs fp64."""
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
B, N, H, D = int(os.environ.get('B', 32)), int(os.environ.get('NQ', 600)), 8, 32
g = torch.Generator(device='cuda').manual_seed(0)
qkv = torch.randn(B, N, 3 * H * D, device='cuda', generator=g)
q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
out = ops.self_attention(q, k, v, H, f16x3=True)
q64, k64, v64 = (x.double().view(B, N, H, D).transpose(1, 2) for x in (q, k, v))
ref = (torch.softmax(q64 @ k64.transpose(-1, -2) / D ** 0.5, -1) @ v64).transpose(1, 2).reshape(B, N, H * D)
err = float((out.double() - ref).abs().max() / ref.abs().max())
print('NW=%s B=%d N=%d: %.4f ms  err %.2e' % (os.environ.get('FF3D_ATTN_NW', 'auto'), B, N, t(lambda: ops.self_attention(q, k, v, H, f16x3=True)), err))
