"""[replay, eager launch, torch.cuda.synchronize(), replay] x N on the captured head at a chosen shape - the sequence runtime.py
refused until round 6 (the memset nodes behind the fault are gone: csrc/heatmap.hip zero_u32).  One run per process.

    python tools/stress_replay_sync.py <form> [--iters N] [--batch B] [--channels C] [--grid G]
      form: graphed    runtime.GraphedHead (one graph, the caller's stream), inputs refilled by copy before every replay
            pipelined  runtime.PipelinedHead, 2 slots / 2 streams, submit() + an eager kernel + a device synchronise per step
            lc         runtime.NeckAndHead (camera maps + LiDAR BEV -> FocalEncoder -> head) captured as one graph

Every replay's packed detections must equal the eager result of the same frames bit for bit; prints RESULT <form> ok."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def opt(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


form = sys.argv[1]
iters, B, C, grid = opt('--iters', 100), opt('--batch', 32), opt('--channels', 256), opt('--grid', 180)
dev = torch.device('cuda', 0)
from focalformer3d_amd import dist as fdist  # noqa: E402
from focalformer3d_amd.runtime import GraphedHead, NeckAndHead, PipelinedHead  # noqa: E402
from focalformer3d_amd.synthetic import (build_head_from_cfg, build_neck_from_cfg, focalformer3d_l_head_cfg,  # noqa: E402
                                         focalformer3d_lc_cfgs, lc_inputs, stage_features)

scratch = torch.zeros(1 << 20, device=dev)
if form == 'lc':
    ncfg, hcfg = focalformer3d_lc_cfgs(C=C)
    neck = build_neck_from_cfg(ncfg, seed=1, device=dev)
    img, pts, metas, _ = lc_inputs(B, seed=2, device=dev)
    metas = [dict(m, box_type_3d=(lambda t, box_dim=9: t)) for m in metas]
    head = NeckAndHead(neck, build_head_from_cfg(hcfg, seed=0, device=dev), metas).eval()
    batches = [[img, [pts]]]
    img2, pts2, _, _ = lc_inputs(B, seed=3, device=dev)
    batches.append([img2, [pts2]])
else:
    head = build_head_from_cfg(focalformer3d_l_head_cfg(C=C, grid=grid, num_proposals=200, stages=3, decoder_stages=2), seed=0, device=dev)
    batches = [stage_features(B, C, grid, 3, seed=11 + i, device=dev) for i in range(2)]


def eager(inputs):
    return fdist.pack_detections(*head.get_bboxes_padded(head(inputs, None, None))).clone()


if form == 'pipelined':
    p = PipelinedHead(head, batches, slots=2)
    want = [p.eager_reference(s).clone() for s in range(2)]
    torch.cuda.synchronize()
    for it in range(iters):
        s = p.submit()
        scratch.add_(1.0)                         # an eager kernel on the caller's stream while the replay runs on the slot's
        torch.cuda.synchronize()                  # the host blocks on the device
        assert torch.equal(p.packed[s], want[s]), ('replay differs from eager launches', it, s)
        if it % 20 == 0:
            print('iter', it, 'ok', flush=True)
else:
    want = [eager(b) for b in batches]
    torch.cuda.synchronize()
    g = GraphedHead(head, batches[0], pack=True)
    for it in range(iters):
        g(batches[it % 2])                        # eager copies into the static inputs, then the replay
        scratch.add_(1.0)
        torch.cuda.synchronize()
        assert torch.equal(g.packed, want[it % 2]), ('replay differs from eager launches', it)
        if it % 20 == 0:
            print('iter', it, 'ok', flush=True)
print('RESULT', form, 'ok', iters, 'x [replay, eager launch, torch.cuda.synchronize()]', flush=True)
