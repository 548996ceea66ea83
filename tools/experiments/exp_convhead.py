import torch, torch.nn.functional as F, sys
sys.path.insert(0, '.')
from focalformer3d_amd import ops
def t(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for B in (8, 32):
    C, K = 256, 10
    x = torch.randn(B, C, 180, 180, device='cuda'); b1 = torch.randn(C, device='cuda')
    w = torch.randn(K, C, 3, 3, device='cuda') * 0.02; b2 = torch.randn(K, device='cuda')
    ms_f = t(lambda: ops.relu_conv3x3_small(x, b1, w, b2))
    ms_v = t(lambda: F.conv2d(ops.bias_relu_(x.clone(), b1), w, b2, padding=1)) - t(lambda: x.clone())
    print(f'B={B}: fused {ms_f:.3f} ms ({2*B*32400*C*K*9/1e9/ms_f:.1f} TF useful) vs vendor bias_relu+conv {ms_v:.3f} ms')
