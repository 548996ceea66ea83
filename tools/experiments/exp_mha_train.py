#!/usr/bin/env python
"""ff3d_mha_train_fwd / _bwd (MFMA or, with FF3D_MHA_TRAIN_SCALAR=1, scalar kernels) against float64: per-tensor errors and times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops  # noqa: E402


def case(B, N, heads, Dh, masked, p_drop, seed=0):
    C_ = heads * Dh
    g = torch.Generator().manual_seed(seed + N + Dh)
    q, k, v, go = (torch.randn(B, N, C_, generator=g) for _ in range(4))
    mask = None
    if masked:
        nq = N - N // 4
        valid = torch.rand(B, N - nq, generator=g) > 0.3
        mask = torch.ones(B, N, N, dtype=torch.bool)
        mask[:, :, :nq] = False
        mask[:, nq:, nq:] = ~(valid[:, None] & valid[:, :, None])
    keep = (torch.rand(B, heads, N, N, generator=g) >= p_drop).to(torch.uint8) if p_drop else None
    ks = 1.0 / (1.0 - p_drop)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    sp = lambda t: t.view(B, N, heads, Dh).transpose(1, 2)
    s_ = sp(qd) @ sp(kd).transpose(-1, -2) / Dh ** 0.5
    if mask is not None:
        s_ = s_.masked_fill(mask[:, None], float('-inf'))
    pr = s_.softmax(-1)
    if keep is not None:
        pr = pr * keep.double() * ks
    ref = (pr @ sp(vd)).transpose(1, 2).reshape(B, N, C_)
    (ref * go.double()).sum().backward()
    dev = 'cuda'
    qc, kc, vc, gc = (t.to(dev) for t in (q, k, v, go))
    m8 = None if mask is None else mask.to(torch.uint8).to(dev)
    kp = None if keep is None else keep.to(dev)
    out, lse = ops.mha_train_fwd(qc, kc, vc, heads, m8, kp, ks)
    gq, gk, gv = ops.mha_train_bwd(qc, kc, vc, heads, out, lse, gc, m8, kp, ks)
    err = lambda a, b: float((a.cpu().double() - b).abs().max() / b.abs().max())
    msg = f'B={B} N={N} heads={heads} Dh={Dh} mask={masked} p={p_drop}: out {err(out, ref.detach()):.1e} dq {err(gq, qd.grad):.1e} ' \
          f'dk {err(gk, kd.grad):.1e} dv {err(gv, vd.grad):.1e}'

    def t_(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e = [torch.cuda.Event(True) for _ in range(2)]
        e[0].record()
        for _ in range(n):
            fn()
        e[1].record()
        torch.cuda.synchronize()
        return e[0].elapsed_time(e[1]) / n * 1e3
    msg += f'   fwd {t_(lambda: ops.mha_train_fwd(qc, kc, vc, heads, m8, kp, ks)):6.1f} us  bwd {t_(lambda: ops.mha_train_bwd(qc, kc, vc, heads, out, lse, gc, m8, kp, ks)):6.1f} us'
    print(msg, flush=True)


if __name__ == '__main__':
    for c in [(1, 200, 2, 64, True, 0.0), (1, 200, 2, 64, False, 0.0), (1, 64, 1, 16, True, 0.0), (1, 16, 1, 16, True, 0.0), (2, 17, 4, 16, True, 0.5),
              (4, 720, 8, 32, True, 0.1), (4, 720, 8, 32, False, 0.0), (4, 720, 8, 32, True, 0.0), (1, 693, 8, 32, True, 0.1)]:
        case(*c)
