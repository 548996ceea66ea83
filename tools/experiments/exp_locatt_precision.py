"""How accurate is the matrix-core local attention on the lc chain's OWN q / k / v (random weights: post-ReLU features, logits of
several hundred), and what does centring the keys buy?  Captures the first block's q, k, v from the scalar path at 1 frame, then
compares against an fp64 evaluation: the scalar fp32 kernel, the MFMA pair kernel as is, the MFMA kernel on keys minus their per-channel
mean (out-of-map pixels = -mean through the planes' zero row: the softmax is shift-invariant per query)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['FF3D_LOCATT_MFMA'] = '0'
from focalformer3d_amd import ops  # noqa: E402
from focalformer3d_amd.synthetic import build_neck_from_cfg, focalformer3d_lc_cfgs, lc_inputs  # noqa: E402

ncfg, hc = focalformer3d_lc_cfgs()
neck = build_neck_from_cfg(ncfg, seed=1, device='cuda')
img, pts, metas, _ = lc_inputs(1, seed=int(os.environ.get('SEED', 3)), device='cuda')
cap = []
orig = ops.local_attention


def spy(q, k, v, ks, scale):
    cap.append((q.clone(), k.clone(), v.clone(), scale))
    return orig(q, k, v, ks, scale)


ops.local_attention = spy
with torch.no_grad():
    neck(img, pts, metas)
ops.local_attention = orig
for blk, (q, k, v, scale) in enumerate(cap):
    B, C, H, W = q.shape
    # fp64 reference by unfold (1 frame)
    qd, kd, vd = q.double(), k.double(), v.double()
    ku = torch.nn.functional.unfold(kd, 9, padding=4).view(B, C, 81, H * W)
    s = (qd.view(B, C, 1, H * W) * ku).sum(1) * scale                                     # (B, 81, HW); out-of-map: 0
    p = torch.softmax(s, 1)
    vu = torch.nn.functional.unfold(vd, 9, padding=4).view(B, C, 81, H * W)
    ref = (vu * p.unsqueeze(1)).sum(2).view(B, C, H, W)
    sc = float(ref.abs().max())
    rows = lambda x: ops.split_f16(x, to_nhwc=True).map(lambda t: t.reshape(B * H * W, C))
    nchw = lambda pr: pr.value().view(B, H, W, C).permute(0, 3, 1, 2).double()
    e_scalar = float((orig(q, k, v, 9, scale).double() - ref).abs().max())
    e_mfma = float((nchw(ops.local_attention_pair(rows(q), rows(k), rows(v), B, H, W, 9, scale)) - ref).abs().max())
    mean = k.mean((2, 3), keepdim=True)
    kc = rows(k - mean)
    # the zero row of the centred planes := -mean (same exponent): split a one-pixel map with the pair's exponent
    m1 = -mean.view(1, C)
    e = 0 if kc.exp is None else int(kc.exp)
    x = torch.ldexp(m1, torch.tensor(-e, device='cuda'))
    hi = x.half()
    lo = ((x - hi.float()) * 2048).half()
    kc[0].view(-1)[B * H * W * C:].copy_(hi.view(-1)) if False else None
    base_hi = torch.as_strided(kc[0], (B * H * W + 1, C), (C, 1))
    base_lo = torch.as_strided(kc[1], (B * H * W + 1, C), (C, 1))
    base_hi[B * H * W].copy_(hi.view(-1))
    base_lo[B * H * W].copy_(lo.view(-1))
    e_cent = float((nchw(ops.local_attention_pair(rows(q), kc, rows(v), B, H, W, 9, scale)) - ref).abs().max())
    sc_c = ((qd.view(B, C, 1, H * W) * (ku - mean.double().view(B, C, 1, 1))).sum(1) * scale)
    print('block %d: |logit| max %.1f (centred %.1f), window spread max %.1f; out scale %.3f; max error vs fp64: scalar fp32 %.2e, '
          'MFMA pairs %.2e, MFMA pairs + centred keys %.2e' % (blk, float(s.abs().max()), float(sc_c.abs().max()),
                                                                 float((s.max(1).values - s.min(1).values).max()), sc, e_scalar, e_mfma, e_cent))
