"""Micro-benchmarks of the dense (vendor-library) pieces of the head, to decide where hand-written work pays."""
import time, torch, torch.nn.functional as F
dev = 'cuda'
def t(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B, C = 8, 256
x = torch.randn(B, C, 180, 180, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.02; b = torch.randn(C, device=dev)
gf = 2 * B * 180 * 180 * C * C * 9 / 1e9
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    ms = t(lambda: F.conv2d(x, w, b, padding=1))
    print(f'conv3x3 {C}->{C} B={B} fp32 NCHW benchmark={bench}: {ms:.3f} ms  {gf/ms:.1f} TFLOP/s')
    xc, wc = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
    ms = t(lambda: F.conv2d(xc, wc, b, padding=1))
    print(f'conv3x3 {C}->{C} B={B} fp32 NHWC benchmark={bench}: {ms:.3f} ms  {gf/ms:.1f} TFLOP/s')
    ms = t(lambda: F.conv2d(x.bfloat16(), w.bfloat16(), b.bfloat16(), padding=1))
    print(f'conv3x3 bf16 (incl casts) benchmark={bench}: {ms:.3f} ms')
    w10 = torch.randn(10, C, 3, 3, device=dev)
    ms = t(lambda: F.conv2d(x, w10, None, padding=1))
    print(f'conv3x3 {C}->10 benchmark={bench}: {ms:.3f} ms')
    ms = t(lambda: F.conv2d(x, w, b, padding=1, stride=2))
    print(f'conv3x3 s2 {C}->{C} benchmark={bench}: {ms:.3f} ms')
torch.backends.cudnn.benchmark = False
M = B * 42525
X = torch.randn(M, C, device=dev)
for N in (256, 768, 1536):
    W = torch.randn(N, C, device=dev); bb = torch.randn(N, device=dev)
    ms = t(lambda: F.linear(X, W, bb))
    print(f'GEMM {M}x{C}x{N} fp32: {ms:.3f} ms {2*M*C*N/1e9/ms:.1f} TFLOP/s')
    Xb, Wb = X.bfloat16(), W.bfloat16()
    ms = t(lambda: F.linear(Xb, Wb))
    print(f'GEMM {M}x{C}x{N} bf16: {ms:.3f} ms {2*M*C*N/1e9/ms:.1f} TFLOP/s')
R = torch.randn(B * 600, 37632, device=dev); W0 = torch.randn(512, 37632, device=dev)
ms = t(lambda: F.linear(R, W0)); print(f'roi_mlp.0 GEMM {B*600}x37632x512 fp32: {ms:.3f} ms {2*B*600*37632*512/1e9/ms:.1f} TFLOP/s')
ms = t(lambda: F.linear(R.bfloat16(), W0.bfloat16())); print(f'roi_mlp.0 bf16 incl cast: {ms:.3f} ms')
q = torch.randn(B, 8, 600, 32, device=dev)
ms = t(lambda: F.scaled_dot_product_attention(q, q, q)); print(f'sdpa fp32 B={B} h=8 N=600 D=32: {ms:.3f} ms')
ms = t(lambda: torch.softmax((q * 32 ** -0.5) @ q.transpose(-1, -2), -1) @ q); print(f'math attention: {ms:.3f} ms')
