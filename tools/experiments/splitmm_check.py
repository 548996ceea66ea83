#!/usr/bin/env python
"""Correctness (vs fp64) and speed (vs MIOpen / hipBLASLt fp32) of the split-fp16 conv / GEMM kernels."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops                                  # noqa: E402


def timed(fn, steps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(steps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / steps


def err(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


def main():
    torch.manual_seed(0)
    dev = 'cuda'
    res = {}
    # ---- correctness: small conv incl. ragged M / N, stride 2, tiny activations (fp16 subnormal range)
    for tag, (B, C, H, W, N, stride, scale) in {'conv_small': (2, 64, 19, 23, 40, 1, 1.0), 'conv_s2': (1, 32, 18, 18, 130, 2, 1.0),
                                                'conv_tiny_values': (1, 32, 16, 16, 16, 1, 1e-4)}.items():
        x = torch.randn(B, C, H, W, device=dev) * scale
        w = torch.randn(N, C, 3, 3, device=dev) * 0.03
        b = torch.randn(N, device=dev)
        ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1)
        out = ops.conv3x3_f16x3(ops.split_f16(x, True), ops.split_weight_f16(w), b, False, stride)
        f32 = F.conv2d(x, w, b, stride=stride, padding=1)
        res[tag] = {'rel_err_f16x3': err(out, ref), 'rel_err_miopen_f32': err(f32, ref)}
    a = torch.randn(300, 96, device=dev)
    w = torch.randn(200, 96, device=dev)
    ref = a.double() @ w.double().t()
    out = ops.gemm_f16x3(ops.split_f16(a), ops.split_weight_f16(w))
    res['gemm_small'] = {'rel_err_f16x3': err(out, ref), 'rel_err_f32': err(a @ w.t(), ref)}
    relu = ops.gemm_f16x3(ops.split_f16(a), ops.split_weight_f16(w), None, True)
    res['gemm_small']['relu_ok'] = bool(torch.equal(relu, out.clamp_min(0)))
    print(json.dumps(res))
    # ---- speed at the head's shapes
    B, C, H = 32, 256, 180
    x = torch.randn(B, C, H, H, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.03
    b = torch.randn(C, device=dev)
    ws = ops.split_weight_f16(w)
    xs = ops.split_f16(x, True)
    out = ops.conv3x3_f16x3(xs, ws, b, True)
    ref = F.relu(F.conv2d(x, w, b, padding=1))
    flop = 2.0 * B * H * H * C * C * 9
    t_split = timed(lambda: ops.split_f16(x, True))
    t_conv = timed(lambda: ops.conv3x3_f16x3(xs, ws, b, True))
    t_ref = timed(lambda: F.conv2d(x, w, None, padding=1))
    sp = {'conv_256_180_b32': {'ms_split': round(t_split, 3), 'ms_f16x3': round(t_conv, 3), 'ms_miopen_f32': round(t_ref, 3),
                               'TF_equiv_f16x3': round(flop / t_conv / 1e9, 1), 'TF_miopen': round(flop / t_ref / 1e9, 1),
                               'max_abs_diff_vs_miopen': float((out - ref).abs().max()), 'ref_absmax': float(ref.abs().max())}}
    del x, xs, out, ref
    M, K, N = 32 * 42525, 256, 256
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    as_, ws = ops.split_f16(a), ops.split_weight_f16(w)
    t_g = timed(lambda: ops.gemm_f16x3(as_, ws))
    t_r = timed(lambda: a @ w.t())
    d = float((ops.gemm_f16x3(as_, ws) - a @ w.t()).abs().max())
    sp['gemm_value_proj'] = {'ms_f16x3': round(t_g, 3), 'ms_f32': round(t_r, 3), 'TF_equiv': round(2.0 * M * K * N / t_g / 1e9, 1),
                             'TF_f32': round(2.0 * M * K * N / t_r / 1e9, 1), 'max_abs_diff': d}
    w3 = torch.randn(768, K, device=dev) * 0.05
    ws3 = ops.split_weight_f16(w3)
    t_g3 = timed(lambda: ops.gemm_f16x3(as_, ws3))
    t_r3 = timed(lambda: a @ w3.t())
    sp['gemm_value_proj_N768'] = {'ms_f16x3': round(t_g3, 3), 'ms_f32': round(t_r3, 3), 'TF_equiv': round(2.0 * M * K * 768 / t_g3 / 1e9, 1),
                                  'max_abs_diff': float((ops.gemm_f16x3(as_, ws3) - a @ w3.t()).abs().max())}
    del a, as_
    M, K, N = 19200, 37632, 512
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.01
    as_, ws = ops.split_f16(a), ops.split_weight_f16(w)
    t_g = timed(lambda: ops.gemm_f16x3(as_, ws))
    t_r = timed(lambda: a @ w.t())
    sp['gemm_roi_mlp0'] = {'ms_f16x3': round(t_g, 3), 'ms_f32': round(t_r, 3), 'TF_equiv': round(2.0 * M * K * N / t_g / 1e9, 1),
                           'TF_f32': round(2.0 * M * K * N / t_r / 1e9, 1)}
    print(json.dumps(sp))


if __name__ == '__main__':
    main()
