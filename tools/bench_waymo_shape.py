#!/usr/bin/env python
"""BASELINE.json configs[4] shape on one MI355X: 468 x 468 BEV (levels 468 / 234 / 117, Nv = 287 469), 1000 queries
(4 HIP stages x 250), K = 3 classes, C = 256, 2 decoder stages x 3 layers, RoI 7x7.  Prints frames/s of
FocalDecoder.forward + get_bboxes_padded for the default fp32-class path and for the bf16 decoder-projection mode."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalformer3d_amd.synthetic import build_head_from_cfg, focalformer3d_l_head_cfg, stage_features   # noqa: E402


def main(B=4, C=256, steps=5):
    dev = 'cuda'
    cfg = focalformer3d_l_head_cfg(C=C, grid=468, num_proposals=250, stages=4, decoder_stages=2, num_classes=3, dataset='Waymo')
    head = build_head_from_cfg(cfg, seed=0, device=dev)
    inputs = stage_features(B, C, 468, 4, seed=1, device=dev)
    metas = [{'box_type_3d': lambda t, box_dim=7: t}] * B
    out = {}
    for tag in ('f32-class', 'bf16 decoder projections'):
        if tag != 'f32-class':
            head.set_gemm_dtype(torch.bfloat16)
        for _ in range(2):
            head.get_bboxes_padded(head(inputs, None, metas))
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record()
        for _ in range(steps):
            head.get_bboxes_padded(head(inputs, None, metas))
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / steps
        out[tag] = {'ms_per_step': round(ms, 3), 'frames_per_s': round(B * 1e3 / ms, 1)}
    print(json.dumps({'workload': f'Waymo shape 468x468x{C}, 1000 queries, B={B}', **out,
                      'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))


if __name__ == '__main__':
    main(B=int(sys.argv[1]) if len(sys.argv) > 1 else 4)
