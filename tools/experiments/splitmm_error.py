#!/usr/bin/env python
"""Error of the split-fp16 GEMM / conv vs fp64, beside the vendor fp32 kernels, at the head's K sizes."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops                                  # noqa: E402


def stats(a, ref):
    d = (a.double() - ref).abs()
    return {'max_rel_to_max': float(d.max() / ref.abs().max()), 'rms_rel': float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())}


def main():
    torch.manual_seed(0)
    dev = 'cuda'
    out = {}
    for tag, (M, K, N, ws) in {'value_proj_K256': (20000, 256, 768, 0.06), 'conv_like_K2304': (20000, 2304, 256, 0.03),
                               'roi_mlp0_K37632': (19200, 37632, 512, 0.007)}.items():
        a = torch.randn(M, K, device=dev).relu_() if 'roi' in tag else torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * ws
        ref = a.double() @ w.double().t()
        mine = ops.gemm_f16x3(ops.split_f16(a), ops.split_weight_f16(w))
        out[tag] = {'f16x3': stats(mine, ref), 'hipblaslt_f32': stats(a @ w.t(), ref)}
        del a, w, ref, mine
    x = torch.randn(4, 256, 180, 180, device=dev)
    w = torch.randn(256, 256, 3, 3, device=dev) * 0.03
    ref = F.conv2d(x.double(), w.double(), padding=1)
    mine = ops.conv3x3_f16x3(ops.split_f16(x, True), ops.split_weight_f16(w))
    out['conv3x3_256'] = {'f16x3': stats(mine, ref), 'miopen_f32': stats(F.conv2d(x, w, padding=1), ref)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
