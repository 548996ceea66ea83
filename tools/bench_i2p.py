#!/usr/bin/env python
"""Camera-projection sampler (I2P, BASELINE configs[2] shape) on one MI355X: 6 x 256 x 232 x 400 camera maps,
180 x 180 BEV pillars, Z = 10 height samples.  Prints one JSON line: module frames/s and the roofline of
cam_sample_kernel (algorithmic bytes = n_valid * 4 corners * Ci * 4 B + qk read + ctx write)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalformer3d_amd import ops                                  # noqa: E402
from focalformer3d_amd.i2p import I2P                              # noqa: E402
from focalformer3d_amd.synthetic import camera_rig                 # noqa: E402
from oracle import ff3d_oracle as O                                # noqa: E402  (only to COUNT visible samples)


def main(B=4, C=256, Ci=256, H=180, W=180, Z=10, Hi=232, Wi=400, steps=10):
    dev = 'cuda'
    torch.manual_seed(0)
    m = I2P(C, Ci, 0.1, max_points_height=Z).eval().to(dev)
    lidar = torch.randn(B, C, H, W, device=dev)
    img = torch.randn(B, 6, Ci, Hi, Wi, device=dev)
    shape = (Hi * 4, Wi * 4)
    l2i = camera_rig(B, 6, shape)
    metas = [dict(lidar2img=l2i[b], input_shape=shape) for b in range(B)]
    n_valid = sum(int(O.i2p_project(torch.from_numpy(l2i[b]), H, W, Z, shape)[1].sum()) for b in range(B))
    for _ in range(3):
        out = m(lidar, img, metas)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(steps):
        out = m(lidar, img, metas)
    e.record()
    torch.cuda.synchronize()
    ms_module = s.elapsed_time(e) / steps
    # kernel alone, same tensors
    img_cl = ops.nchw_to_nhwc(img.view(B * 6, Ci, Hi, Wi)).view(B, 6, Hi, Wi, Ci)
    qk = torch.randn(B, H * W, Ci, device=dev)
    l2i_t = torch.from_numpy(l2i).to(dev)
    args = (img_cl, l2i_t, None, qk, H, W, Z, (-54.0, -54.0, -5.0, 54.0, 54.0, 3.0), shape)
    for _ in range(3):
        ops.cam_sample(*args)
    s.record()
    for _ in range(steps):
        ops.cam_sample(*args)
    e.record()
    torch.cuda.synchronize()
    ms_k = s.elapsed_time(e) / steps
    flat = img.view(B * 6, Ci, Hi, Wi)
    for _ in range(3):
        ops.nchw_to_nhwc(flat)
    torch.cuda.synchronize()
    s.record()
    for _ in range(steps):
        ops.nchw_to_nhwc(flat)
    e.record()
    torch.cuda.synchronize()
    ms_t = s.elapsed_time(e) / steps
    alg = n_valid * 4 * Ci * 4 + 2 * B * H * W * Ci * 4
    tr_bytes = 2 * B * 6 * Ci * Hi * Wi * 4
    print(json.dumps({
        'workload': f'I2P.forward, {B} frames, 6x{Ci}x{Hi}x{Wi} camera maps, {H}x{W}x{C} BEV, Z={Z} (BASELINE configs[2])',
        'frames_per_s': round(B / ms_module * 1e3, 2), 'ms_per_call': round(ms_module, 3),
        'visible_fraction_of_pillars': round(float((out.abs().sum(1) > 0).float().mean()), 4),
        'n_valid_point_camera_pairs': n_valid,
        'cam_sample_kernel': {'ms': round(ms_k, 4), 'algorithmic_bytes': alg, 'achieved_GBs': round(alg / ms_k / 1e6, 1),
                              'frac_of_8TBs': round(alg / ms_k / 1e6 / 8000, 4)},
        'nchw_to_nhwc_kernel': {'ms': round(ms_t, 4), 'bytes': tr_bytes, 'achieved_GBs': round(tr_bytes / ms_t / 1e6, 1),
                                'frac_of_8TBs': round(tr_bytes / ms_t / 1e6 / 8000, 4)}}))


if __name__ == '__main__':
    main(B=int(sys.argv[1]) if len(sys.argv) > 1 else 4)
