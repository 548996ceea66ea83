"""Times the feed-forward step of a decoder layer at a given row count: ops.ffn_rows (one launch, csrc/ffnrows.hip) against the two-launch
form (linear.hip fc1 + ReLU, linrows.hip fc2 + add + LayerNorm).  FF3D_FFN_MT=2..5 forces the block height of the fused kernel.
usage: python tools/bench_ffn_rows.py [rows] [hidden] [fused|two-launch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalformer3d_amd import ops  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 19200
    hidden = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, 256, generator=g).to(dev)
    w1, b1 = (torch.randn(hidden, 256, generator=g) * 0.05).to(dev), torch.randn(hidden, generator=g).to(dev)
    w2, b2 = (torch.randn(256, hidden, generator=g) * 0.05).to(dev), torch.randn(256, generator=g).to(dev)
    pos = torch.randn(M, 256, generator=g).to(dev)
    gamma, beta = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    w1t, w2t = ops.tile_weight_f16(w1, bias=b1), ops.tile_weight_f16(w2, bias=b2)
    s1, s2 = ops.split_weight_f16(w1, bias=b1), ops.split_weight_f16(w2, bias=b2)

    def fused():
        return ops.ffn_rows(x, w1t, b1, w2t, b2, x, gamma, beta, 1e-5, pos)

    def two():
        return ops.linear_rows(ops.linear_f16x3(x, s1, b1, True), s2, b2, residual=x, gamma=gamma, beta=beta, eps=1e-5, pos=pos)

    only = sys.argv[3] if len(sys.argv) > 3 else None
    for name, fn in (('fused', fused), ('two-launch', two)):
        if only and name != only:
            continue
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f'{name:11s} rows {M} hidden {hidden} FF3D_FFN_MT={os.environ.get("FF3D_FFN_MT", "auto")}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us')


if __name__ == '__main__':
    main()
