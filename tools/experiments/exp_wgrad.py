#!/usr/bin/env python
"""ff3d_linear_wgrad_f16x3 against fp64 and against the framework's fp32 GEMM (dy^T @ x + column sums): error and time per shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


def main():
    torch.manual_seed(0)
    shapes = [(170100, 256, 256), (170100, 256, 64), (2880, 256, 1024), (2880, 1024, 256), (2880, 256, 192), (2880, 256, 96),
              (19200 // 8, 37632, 512), (1001, 36, 20), (31, 8, 4), (4 * 32400, 256, 12)]
    if os.environ.get('ONLY_FIRST') == '1':
        shapes = shapes[:1]
    if os.environ.get('TIME_ONLY') == '1':              # ablation runs (experiments library): results are wrong by design
        M, K, N = shapes[0]
        x = torch.randn(M, K, device='cuda')
        dy = torch.randn(M, N, device='cuda')
        print(f"FF3D_WG_ABLATE={os.environ.get('FF3D_WG_ABLATE', '0')}: {timeit(lambda: ops.linear_wgrad(x, dy), 50):8.1f} us for the 4 launches")
        return
    for M, K, N in shapes:
        x = torch.randn(M, K, device='cuda') * 3.0
        dy = torch.randn(M, N, device='cuda') * 1e-4 * torch.rand(M, 1, device='cuda') ** 4
        dw, db = ops.linear_wgrad(x, dy)
        ref = dy.double().t() @ x.double()
        refb = dy.double().sum(0)
        scale = (dy.double().abs().t() @ x.double().abs()).max()       # sum of |products|: the natural unit of a dot product's error
        e_own = float((dw.double() - ref).abs().max() / scale)
        v = dy.t() @ x
        e_ven = float((v.double() - ref).abs().max() / scale)
        eb = float((db.double() - refb).abs().max() / dy.double().abs().sum(0).max())
        t_own = timeit(lambda: ops.linear_wgrad(x, dy))
        t_ven = timeit(lambda: (dy.t() @ x, dy.sum(0)))
        print(f'M={M:7d} K={K:6d} N={N:5d}  err/sum|prod| own {e_own:.2e} vendor-fp32 {e_ven:.2e}  bias {eb:.1e}   '
              f'own {t_own:8.1f} us (4 launches)  vendor {t_ven:8.1f} us', flush=True)
    # column-block operands (row-strided views)
    big = torch.randn(5000, 512, device='cuda')
    x, dy = big[:, 128:384], big[:, 384:512] * 1e-3
    dw, db = ops.linear_wgrad(x, dy)
    ref = dy.double().t() @ x.double()
    print('strided views: err', float((dw.double() - ref).abs().max() / (dy.double().abs().t() @ x.double().abs()).max()))


if __name__ == '__main__':
    main()
