"""Second round of dense micro-benchmarks: NHWC variants, small convs, unfold+GEMM, transposes."""
import torch, torch.nn.functional as F
dev = 'cuda'
def t(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B, C = 8, 256
CL = torch.channels_last
for (H, stride, Cout, name) in [(180, 1, 256, 'hm conv1'), (180, 1, 10, 'hm conv2'), (180, 2, 256, 'dconv'), (90, 2, 256, 'dconv2')]:
    x = torch.randn(B, C, H, H, device=dev); w = torch.randn(Cout, C, 3, 3, device=dev) * 0.02; b = torch.randn(Cout, device=dev)
    gf = 2 * B * (H // stride) ** 2 * C * Cout * 9 / 1e9
    ms = t(lambda: F.conv2d(x, w, b, padding=1, stride=stride)); print(f'{name} NCHW: {ms:.3f} ms {gf/ms:.1f} TF')
    xc, wc = x.contiguous(memory_format=CL), w.contiguous(memory_format=CL)
    ms = t(lambda: F.conv2d(xc, wc, b, padding=1, stride=stride)); print(f'{name} NHWC: {ms:.3f} ms {gf/ms:.1f} TF')
    ms = t(lambda: x.contiguous(memory_format=CL)); print(f'   NCHW->NHWC torch copy: {ms:.3f} ms')
    if stride == 2 or Cout == 10:
        wm = w.view(Cout, -1)
        def unf():
            cols = F.unfold(x, 3, padding=1, stride=stride)            # (B, C*9, L)
            return torch.matmul(wm, cols) + b[:, None]
        ms = t(unf); print(f'{name} unfold+matmul: {ms:.3f} ms {gf/ms:.1f} TF')
# elementwise references
x = torch.randn(B, 600, C, device=dev); y = torch.randn_like(x); g = torch.ones(C, device=dev)
ms = t(lambda: F.layer_norm(x + y, (C,), g, g)); print(f'add+LN (B*600 x {C}): {ms*1e3:.1f} us')
big = torch.randn(B, C, 180, 180, device=dev)
ms = t(lambda: F.relu_(big)); print(f'relu_ on (B,C,180,180): {ms*1e3:.1f} us')
