"""heatmap_nms micro-benchmark: B frames x K classes x H x W (default 32 x 10 x 180 x 180 = one stage of the benchmarked step);
algorithmic bytes = logits R + mask R + heat W + mask clone W."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops


def t(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for B, K, H, W in ((32, 10, 180, 180), (4, 10, 180, 180), (8, 3, 468, 468)):
    g = torch.Generator(device='cuda').manual_seed(0)
    logits = torch.randn(B, K, H, W, device='cuda', generator=g) * 2
    mask = (torch.rand(B, K, H, W, device='cuda', generator=g) > 0.2).float()
    bits = ops.small_class_bits('nuScenes' if K == 10 else 'Waymo', K)
    us = t(lambda: ops.heatmap_nms(logits, mask, None, 3, bits))
    nbytes = 4 * logits.numel() * 4
    print(f'B={B} K={K} {H}x{W}: {us:.1f} us  {nbytes / 1e6:.0f} MB  {nbytes / us / 1e6:.2f} TB/s = {nbytes / us / 1e6 / 8:.3f} of the 8 TB/s peak '
          f'(includes the histogram memset launch)')
