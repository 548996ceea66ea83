"""Where do the two local-attention routes of the 'bevfusion' block part ways?  Same neck, same inputs, FF3D_LOCATT_MFMA on / off in
one process; per block: q / k / v (pair value vs fp32 rows), context, block output."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import focal_encoder as FE, ops  # noqa: E402
from focalformer3d_amd.synthetic import build_neck_from_cfg, focalformer3d_lc_cfgs, lc_inputs  # noqa: E402

ncfg, hc = focalformer3d_lc_cfgs()
neck = build_neck_from_cfg(ncfg, seed=1, device='cuda')
img, pts, metas, _ = lc_inputs(1, seed=3, device='cuda')
rec = {0: [], 1: []}
o_scalar, o_pair = ops.local_attention, ops.local_attention_pair


def spy_scalar(q, k, v, ks, scale):
    out = o_scalar(q, k, v, ks, scale)
    rec[0].append([t.clone() for t in (q, k, v, out)])
    return out


def spy_pair(q, k, v, B, H, W, ks, scale):
    out = o_pair(q, k, v, B, H, W, ks, scale)
    nchw = lambda pr: pr.value().view(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
    rec[1].append([nchw(ops.as_pair(t)) for t in (q, k, v, out)] + [[None if ops.as_pair(t).exp is None else int(ops.as_pair(t).exp) for t in (q, k, v)]])
    return out


ops.local_attention, ops.local_attention_pair = spy_scalar, spy_pair
outs = {}
with torch.no_grad():
    for mode in (0, 1):
        FE.LOCATT_MFMA = bool(mode)
        outs[mode] = neck(img, pts, metas)[1]
maps = lambda o: [o[0]] + list(o[1])
for i, (a, b) in enumerate(zip(maps(outs[0]), maps(outs[1]))):
    print('map %d: max |scalar route - MFMA route| %.3e (scale %.3f)' % (i, float((a - b).abs().max()), float(a.abs().max())))
for blk, (s, m) in enumerate(zip(rec[0], rec[1])):
    names = ('q', 'k', 'v', 'context')
    print('block', blk, 'pair exponents q k v:', m[4], ' '.join('%s: max diff %.3e (scale %.3f)' % (n, float((x - y).abs().max()), float(x.abs().max()))
                                                                for n, x, y in zip(names, s[:4], m[:4])))
    # the MFMA kernel on the SCALAR route's fp32 q / k / v (isolates the kernel from its inputs)
    q, k, v, ctx = s
    B, C, H, W = q.shape
    rows = lambda x: ops.split_f16(x, to_nhwc=True).map(lambda t: t.reshape(B * H * W, C))
    alt = o_pair(rows(q), rows(k), rows(v), B, H, W, 9, C ** -0.5).value().view(B, H, W, C).permute(0, 3, 1, 2)
    d = (alt - ctx).abs()
    print('   MFMA kernel on the scalar route\'s own q / k / v vs the scalar kernel: max %.3e at %s' % (float(d.max()), tuple(int(i) for i in torch.nonzero(d == d.max())[0])))
