#!/usr/bin/env python
"""Host-side cost of a training step of the FocalFormer3D-L head (tools/bench_train_step.py's step): cProfile of forward, of
targets + loss and of backward separately, top functions by own and by cumulative time.  The GPU is NOT synchronised inside the
phases: what is listed is what the host spends launching.   python tools/profile_host_train.py [C] [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalformer3d_amd.synthetic import build_head_from_cfg, focalformer3d_l_head_cfg, stage_features   # noqa: E402


def main(C=256, steps=6, B=4, n_gt=40):
    cfg = focalformer3d_l_head_cfg(C=C, grid=180, num_proposals=200, stages=3, decoder_stages=2)
    cfg['train_cfg'] = dict(
        dataset='nuScenes',
        assigner=dict(type='HungarianAssigner3D', iou_calculator=dict(type='BboxOverlaps3D', coordinate='lidar'),
                      cls_cost=dict(type='FocalLossCost', gamma=2, alpha=0.25, weight=0.15),
                      reg_cost=dict(type='BBoxBEVL1Cost', weight=0.25), iou_cost=dict(type='IoU3DCost', weight=0.25)),
        pos_weight=-1, gaussian_overlap=0.1, min_radius=2, grid_size=[1440, 1440, 40], voxel_size=[0.075, 0.075, 0.2],
        out_size_factor=8, code_weights=[1.0] * 8 + [0.2, 0.2], point_cloud_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0])
    head = build_head_from_cfg(cfg, seed=0, device='cuda').train()
    inputs = stage_features(B, C, 180, 3, seed=1, device='cuda')
    g = torch.Generator().manual_seed(2)
    gts, labels = [], []
    for b in range(B):
        t = torch.zeros(n_gt, 9)
        t[:, :2] = torch.rand(n_gt, 2, generator=g) * 100 - 50
        t[:, 2] = torch.rand(n_gt, generator=g) * 2 - 2.5
        t[:, 3:6] = torch.rand(n_gt, 3, generator=g) * torch.tensor([2.0, 4.0, 1.5]) + torch.tensor([0.6, 0.8, 1.0])
        t[:, 6] = (torch.rand(n_gt, generator=g) - 0.5) * 6.2
        gts.append(t.cuda())
        labels.append(torch.randint(0, 10, (n_gt,), generator=g).cuda())
    opt = torch.optim.AdamW(head.parameters(), lr=1e-4, weight_decay=0.01)
    prof = {k: cProfile.Profile() for k in ('forward', 'targets+loss', 'backward')}
    wall = dict.fromkeys(prof, 0.0)
    for it in range(steps + 3):
        on = it >= 3
        opt.zero_grad()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if on: prof['forward'].enable()
        preds = head(inputs, None, [{}] * B, gt_bboxes_3d=gts, gt_labels_3d=labels)
        if on: prof['forward'].disable()
        t1 = time.perf_counter()
        if on: prof['targets+loss'].enable()
        losses = head.loss(gts, labels, preds)
        total = sum(v for n, v in losses.items() if 'loss' in n)
        if on: prof['targets+loss'].disable()
        t2 = time.perf_counter()
        if on: prof['backward'].enable()
        total.backward()
        if on: prof['backward'].disable()
        t3 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize()
        if on:
            wall['forward'] += (t1 - t0) / steps * 1e3
            wall['targets+loss'] += (t2 - t1) / steps * 1e3
            wall['backward'] += (t3 - t2) / steps * 1e3
    for k, pr in prof.items():
        print(f'==== {k}: host {wall[k]:.2f} ms per step (launch side only, under cProfile)')
        for key, n in (('tottime', 28), ('cumulative', 28)):
            buf = io.StringIO()
            pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(n)
            lines = [l for l in buf.getvalue().splitlines() if l.strip()]
            print('\n'.join(l[:170] for l in lines[3:]))


if __name__ == '__main__':
    main(*(int(v) for v in sys.argv[1:3]))
