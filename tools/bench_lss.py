#!/usr/bin/env python
"""Lift-Splat-Shoot camera branch (DeformFormer3D_C_R50.py shape) on one MI355X: 6 x 256 x 112 x 200 camera maps,
41 depth bins, 180 x 180 x 13 voxels of 0.6 m, camC = 64.  Prints one JSON line: module frames/s and the stage split
(cell table = geometry/binning kernel + key sort + offsets | NHWC | depth-net GEMM + softmax | fused lift-splat | the BEV encoder's
four convs timed on MIOpen fp32 for reference - the module itself runs them on the split-fp16 kernels)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalformer3d_amd import ops                                  # noqa: E402
from focalformer3d_amd.lss import LiftSplatShoot                   # noqa: E402
from focalformer3d_amd.synthetic import camera_rig, randomize_     # noqa: E402


def timed(fn, steps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(steps):
        r = fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / steps, r


def main(B=1, steps=5):
    dev = 'cuda'
    torch.manual_seed(0)
    scale = (448, 800)
    m = randomize_(LiftSplatShoot(img_scale=scale, pc_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], downsample=4, grid=0.6,
                                  inputC=256, outputC=128, camC=64), 0).eval().to(dev)
    x = torch.randn(B, 6, 256, 112, 200, device=dev)
    inv = torch.inverse(torch.from_numpy(camera_rig(B, 6, scale))).to(dev)
    rots, trans = inv[..., :3, :3].contiguous(), inv[..., :3, 3].contiguous()
    metas = [{} for _ in range(B)]
    ms_module, (bev, depth) = timed(lambda: m(x, rots, trans, img_metas=metas), steps)
    with torch.no_grad():
        ms_geom, (src, offsets, n_cells) = timed(lambda: m.cell_table(rots, trans), steps)
        axes = [m.frustum[0, 0, :, 0].contiguous(), m.frustum[0, :, 0, 1].contiguous(), m.frustum[:, 0, 0, 2].contiguous()]
        ms_keys, keys = timed(lambda: ops.lss_cells(rots, trans, *axes, (m.bx - m.dx / 2).tolist(), m.dx.tolist(),
                                                    [int(v) for v in m.nx]), steps)
        ms_sort, _ = timed(lambda: torch.sort(keys, stable=True), steps)
        lengths = offsets[1:] - offsets[:-1]
        n_e = int(offsets[n_cells])
        occupied = int((lengths > 0).sum())
        P, D = B * 6 * 112 * 200, m.D
        y = torch.randn(P, 108, device=dev)
        dp = torch.softmax(torch.randn(P, D, device=dev), 1)
        ms_splat, vox = timed(lambda: ops.lss_splat(y[:, :64], dp, src, offsets, n_cells), steps)
        xcl = torch.randn(P, 256, device=dev)
        w = torch.randn(108, 256, device=dev)
        ms_gemm, _ = timed(lambda: torch.softmax(torch.nn.functional.linear(xcl, w)[:, 64:105], 1).contiguous(), steps)
        ms_tr, _ = timed(lambda: ops.nchw_to_nhwc(x.view(B * 6, 256, 112, 200)), steps)
        bevin = torch.randn(B, 832, 180, 180, device=dev)

        def enc():
            t = bevin
            mods = list(m.bevencode)
            for i in range(0, len(mods), 3):
                t = ops.bias_relu_(torch.nn.functional.conv2d(t, mods[i].weight, None, padding=1), mods[i + 1].bias)
            return t
        ms_enc, _ = timed(enc, steps)
    alg = n_e * 8 + P * 64 * 4 + n_cells * (4 + 64 * 4)                    # entries (src + depth) + feature rows once + output
    print(json.dumps({'metric': 'lss_frames_per_s', 'value': round(B * 1e3 / ms_module, 2), 'ms_module': round(ms_module, 3),
                      'ms_cell_table': round(ms_geom, 3), 'ms_cells_kernel': round(ms_keys, 4), 'ms_key_sort': round(ms_sort, 3),
                      'ms_nhwc': round(ms_tr, 3), 'ms_depthnet_softmax': round(ms_gemm, 3),
                      'ms_splat_kernel': round(ms_splat, 4),
                      'ms_bev_encoder_vendor_fp32': round(ms_enc, 3), 'entries': n_e, 'kept_frac': round(n_e / (P * D), 3),
                      'occupied_cells': occupied, 'mean_interval': round(n_e / max(occupied, 1), 1),
                      'max_interval': int(lengths.max()),
                      'splat_alg_GBps': round(alg / ms_splat / 1e6, 1), 'B': B}))


if __name__ == '__main__':
    main(B=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
