"""value_proj GEMM micro-benchmark: M = B * 42525, K = 256, N = 768 (+ bias), fp32 out.  FF3D_GEMM_WS=0|1 per process."""
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


B = int(os.environ.get('B', 32)); K = int(os.environ.get('K', 256)); N = int(os.environ.get('N', 768))
M = B * 42525
g = torch.Generator(device='cuda').manual_seed(0)
a = torch.randn(M, K, device='cuda', generator=g)
w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
b = torch.randn(N, device='cuda', generator=g)
asp, wsp = ops.split_f16(a), ops.split_weight_f16(w, bias=b)
out = ops.gemm_f16x3(asp, wsp, b)
err = 0.0
for lo in (0, M // 2 - 1000, M - 4096):
    ref = a[lo:lo + 4096].double() @ w.double().t() + b.double()
    err = max(err, float((out[lo:lo + 4096].double() - ref).abs().max() / ref.abs().max()))
ms = t(lambda: ops.gemm_f16x3(asp, wsp, b))
print('WS=%s M=%d K=%d N=%d  %.3f ms  (%.0f TF fp16-pass, out %.2f TB/s)  err %.2e' % (
    os.environ.get('FF3D_GEMM_WS', 'default'), M, K, N, ms, 6.0 * M * N * K / 1e9 / ms, M * N * 4 / 1e9 / ms, err))
