#!/usr/bin/env python
"""FocalEncoder neck (FocalFormer3D_L-like: bevfusionmb2, LiDAR only, 2 blocks, extra map) feeding the head's shape on one
MI355X: frames/s of the neck alone and of neck -> head -> get_bboxes.  FF3D_DENSE_MODE=vendor for the MIOpen fp32 convs."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalformer3d_amd.focal_encoder import NECKS                                        # noqa: E402
from focalformer3d_amd.synthetic import build_head_from_cfg, focalformer3d_l_head_cfg, randomize_   # noqa: E402


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


def main(B=8, C=256):
    dev = 'cuda'
    ncfg = dict(num_layers=2, in_channels_img=64, in_channels_pts=384, hidden_channel=C, iterbev='bevfusionmb2',
                max_points_height=5, multistage_heatmap=2, input_img=False, input_pts=True, iterbev_wo_img=True,
                extra_feat=True, iter_bev_cam=False, cam_lss=False)
    neck = randomize_(NECKS.build(dict(ncfg, type='FocalEncoder')), 1).eval().to(dev)
    head = build_head_from_cfg(focalformer3d_l_head_cfg(C=C, grid=180, num_proposals=200, stages=3, decoder_stages=2),
                               seed=0, device=dev)
    pts = torch.randn(B, 384, 180, 180, device=dev)
    metas = [{'box_type_3d': lambda t, box_dim=9: t}] * B
    ms_neck = timed(lambda: neck(None, pts, metas))

    def chain():
        _, inputs = neck(None, pts, metas)
        return head.get_bboxes_padded(head(inputs, None, metas))
    ms_chain = timed(chain)
    print(json.dumps({'B': B, 'C': C, 'dense_mode': os.environ.get('FF3D_DENSE_MODE', 'f16x3'),
                      'neck_ms': round(ms_neck, 3), 'neck_frames_per_s': round(B * 1e3 / ms_neck, 1),
                      'neck_head_ms': round(ms_chain, 3), 'neck_head_frames_per_s': round(B * 1e3 / ms_chain, 1)}))


if __name__ == '__main__':
    main(B=int(sys.argv[1]) if len(sys.argv) > 1 else 8)
