#!/usr/bin/env python
"""One training step of the FocalFormer3D-L head on one MI355X: forward with ground truth (600 queries +
3 ground-truth groups, batch-statistics BatchNorm, dropout), Hungarian targets + losses, backward, AdamW step.  Prints the
per-phase milliseconds (HIP events).  Synthetic maps / boxes; C = 128 (REF) or 256."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalformer3d_amd.synthetic import build_head_from_cfg, focalformer3d_l_head_cfg, stage_features   # noqa: E402


def main(B=4, C=128, steps=5, n_gt=40):
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
    names = ('forward', 'targets+loss', 'backward', 'optimizer')
    acc = dict.fromkeys(names, 0.0)
    for it in range(steps + 2):
        ev = [torch.cuda.Event(True) for _ in range(5)]
        opt.zero_grad()
        ev[0].record()
        preds = head(inputs, None, [{}] * B, gt_bboxes_3d=gts, gt_labels_3d=labels)
        ev[1].record()
        losses = head.loss(gts, labels, preds)
        total = sum(v for n, v in losses.items() if 'loss' in n)
        ev[2].record()
        total.backward()
        ev[3].record()
        opt.step()
        ev[4].record()
        torch.cuda.synchronize()
        if it >= 2:
            for i, n in enumerate(names):
                acc[n] += ev[i].elapsed_time(ev[i + 1]) / steps
    ms = sum(acc.values())
    print(json.dumps({'workload': f'FocalFormer3D-L head training step, C={C}, B={B}, {n_gt} gt boxes/frame, 600+{3 * n_gt} queries',
                      'ms_per_step': round(ms, 2), 'frames_per_s': round(B * 1e3 / ms, 1),
                      'phases_ms': {n: round(v, 2) for n, v in acc.items()}, 'loss': round(float(total), 4),
                      'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == '__main__':
    main(B=int(sys.argv[1]) if len(sys.argv) > 1 else 4, C=int(sys.argv[2]) if len(sys.argv) > 2 else 128)
