#!/usr/bin/env python
"""Host-side cost of one eager decoder step (small batch is host-bound: ~155 launches per step): cProfile over N steps of
``head.forward`` + ``get_bboxes_padded`` at batch B, top functions by own time.   python tools/profile_host.py [B] [steps]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalformer3d_amd.synthetic import build_head_from_cfg, focalformer3d_l_head_cfg, stage_features   # noqa: E402


def main(B=4, steps=30):
    head = build_head_from_cfg(focalformer3d_l_head_cfg(C=256, grid=180, num_proposals=200, stages=3, decoder_stages=2), seed=0,
                               device='cuda')
    inputs = stage_features(B, 256, 180, 3, seed=1, device='cuda')
    metas = [{}] * B
    for _ in range(5):
        head.get_bboxes_padded(head(inputs, None, metas))
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        head.get_bboxes_padded(head(inputs, None, metas))
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(45)
    total = sum(v[2] for v in st.stats.values())
    print(f'host time per step (profiled): {total / steps * 1e3:.3f} ms')


if __name__ == '__main__':
    main(*(int(v) for v in sys.argv[1:3]))
