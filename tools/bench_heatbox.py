"""Cost of the heatmap_box branch (heatmap_box + thin_heatmap_box [+ mask_heatmap_mode='boxcls'], FD:231-287 / 708-782) on the
benchmarked step: FocalFormer3D_L head, 180 x 180 x 256, 3 x 200 queries, eager launches, B frames - ms per step with the branch
off / on / on + boxcls, and the time of its own launches (task-head convs, box gather, box mask) from HIP events.
    python tools/bench_heatbox.py [--batch 32] [--steps 10]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalformer3d_amd import ops  # noqa: E402
from focalformer3d_amd.synthetic import build_head_from_cfg, focalformer3d_l_head_cfg, stage_features  # noqa: E402


def timed(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--steps', type=int, default=10)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    inputs = stage_features(a.batch, 256, 180, 3, seed=1, device=dev)
    rec = {'batch': a.batch, 'steps': a.steps}
    for name, kw in (('off', {}), ('heatmap_box', dict(heatmap_box=True, thin_heatmap_box=True)),
                     ('heatmap_box+boxcls', dict(heatmap_box=True, thin_heatmap_box=True, mask_heatmap_mode='boxcls'))):
        cfg = focalformer3d_l_head_cfg(C=256, grid=180, num_proposals=200, stages=3, decoder_stages=2)
        cfg.update(kw)
        head = build_head_from_cfg(cfg, seed=0, device=dev)
        if kw:
            with torch.no_grad():
                for m in head.multi_stage_task_heads:           # boxes of a few metres
                    m[1].bias.add_(1.0)
            head.invalidate_cache()

        def step():
            return head.get_bboxes_padded(head(inputs, None, None))
        ms = timed(step, a.steps)
        rec[name] = {'ms_per_step': round(ms, 3), 'frames_per_s': round(a.batch / ms * 1e3, 1)}
        if kw:
            out = head(inputs, None, None)[0][0]
            m = out['multistage_masks']
            rec[name]['blanked_cells_per_frame_after_stage_2'] = round(float((m[2] == 0).sum()) / a.batch, 1)
            # the branch's own launches, one by one
            d = head._derived()
            x = inputs[1][0]
            t_head = timed(lambda: head._task_head(x, 1, d), a.steps)
            raw = head._task_head(x, 1, d)
            idx = torch.stack([torch.randperm(10 * 180 * 180, device=dev)[:200] for _ in range(a.batch)])
            qb = torch.zeros(a.batch, 10, 600, device=dev)
            t_g = timed(lambda: ops.heatmap_box_gather(raw, idx, qb, 200, 10), a.steps)
            lab = torch.randint(0, 10, (a.batch, 600), device=dev)
            mask = torch.ones(a.batch, 10, 180, 180, device=dev)
            coder = head.bbox_coder.coder_params
            t_m = timed(lambda: ops.box_class_mask(qb, lab, mask, 200, 200, coder, (-54.0, -54.0, 54.0, 54.0), 3,
                                                   ops.small_class_bits('nuScenes', 10)), a.steps)
            rec[name]['launches_ms'] = {'task head (halo conv + 4 tail launches + cat)': round(t_head, 4),
                                        'ff3d_heatmap_box_gather': round(t_g, 4), 'ff3d_box_class_mask': round(t_m, 4)}
        del head
        torch.cuda.empty_cache()
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
