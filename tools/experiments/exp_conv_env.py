import os, sys, torch, torch.nn.functional as F
def t(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
torch.backends.cudnn.benchmark = os.environ.get('BENCHMARK', '0') == '1'
B, C = 8, 256
x = torch.randn(B, C, 180, 180, device='cuda'); w = torch.randn(C, C, 3, 3, device='cuda') * 0.02
ms = t(lambda: F.conv2d(x, w, None, padding=1))
print(sys.argv[1:], f'conv1 {ms:.3f} ms {2*B*32400*C*C*9/1e9/ms:.1f} TF')
