"""MSDA gather micro-benchmark at the benchmarked shape (B x 600 queries x 8 heads x 3 levels x 4 points, Dh = 32, fp32 value as a column
block of the stage GEMM's output), fused form (softmax + reference-point arithmetic inside).  A/B: FF3D_MSDA_PT4=0 | 1."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops


def t(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for B in (32, 4):
    Nq, M, L, P, C = 600, 8, 3, 4, 256
    hw = [(180, 180), (90, 90), (45, 45)]
    Nv = sum(h * w for h, w in hw)
    g = torch.Generator(device='cuda').manual_seed(B)
    wide = torch.randn(B, Nv, 3 * C, device='cuda', generator=g)
    v = wide[:, :, C:2 * C].unflatten(2, (M, C // M))                     # the middle layer's column block (cell stride 768)
    ref = torch.rand(B, Nq, 2, device='cuda', generator=g)
    both = torch.randn(B * Nq, M * L * P * 3, device='cuda', generator=g)
    both[:, :M * L * P * 2] *= 3.0                                        # offsets of a few cells
    n_off = M * L * P * 2
    out = torch.empty(B, Nq, C, device='cuda')
    us = t(lambda: ops.msda_fused_fwd(v, hw, ref, both[:, :n_off], both[:, n_off:], P, out))
    alg = ops.msda_algorithmic_bytes(B, Nq, M, C // M, L, P, 4)
    print(f'PT4={os.environ.get("FF3D_MSDA_PT4", "1")} B={B}: {us:.1f} us  {alg / 1e6:.0f} MB algorithmic = {alg / us / 1e6:.2f} TB/s = {alg / us / 1e6 / 8:.3f} of 8 TB/s')
