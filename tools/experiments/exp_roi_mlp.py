"""Round 4 A/B (VERDICT r03 "next" #7): roi_mlp.0 of the 32-frame step (M = 19 200 rows, K = 37 632, N = 512) on the shipped
128 x 128 tile kernel (splitmm_kernel<2, 2>, two blocks per CU) for every split-K count, and - when the experiments library is
loaded with FF3D_SPLITMM_VARIANT=4 - on the 256 x 128 / 8-wave / triple-buffered instance (<4, 3>: one block per CU, 25 % fewer
LDS-DMA bytes per MFMA; a 128 x 256 tile has the same bytes per FLOP, so this instance stands in for "256-wide N tiles").
Prints one JSON line; the result of split-K = 1 on the loaded kernel is checked against an fp64 product of a row sample."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops  # noqa: E402


def t(fn, n=8, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    M, K, N = 19200, 37632, 512
    g = torch.Generator(device='cuda').manual_seed(3)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) * 0.01
    b = torch.randn(N, device='cuda', generator=g)
    a_s, w_s = ops.split_f16(a), ops.split_weight_f16(w)
    rows = torch.arange(0, M, 997, device='cuda')
    want = (a[rows].double() @ w.double().t() + b.double()).relu()
    res = {'variant': os.environ.get('FF3D_SPLITMM_VARIANT', 'default'), 'lib': os.path.basename(os.environ.get('FF3D_LIB', 'libff3d_hip.so')),
           'auto_ksplit': ops.gemm_ksplit(M, N, K), 'ms': {}}
    for ks in (1, 2, 3, 4, 5, 6, 7, 8):
        out = ops.gemm_f16x3(a_s, w_s, b, relu=True, ksplit=ks)
        err = float(((out[rows].double() - want).abs().max() / want.abs().max()))
        assert err < 1e-5, (ks, err)
        res['ms'][ks] = round(t(lambda: ops.gemm_f16x3(a_s, w_s, b, relu=True, ksplit=ks)), 4)
        res['max_rel_err_vs_fp64'] = max(res.get('max_rel_err_vs_fp64', 0.0), err)
    best = min(res['ms'], key=res['ms'].get)
    res['best'] = {'ksplit': best, 'ms': res['ms'][best], 'fp16_mfma_tflops': round(3 * 2.0 * M * N * K / res['ms'][best] / 1e9, 1)}
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
