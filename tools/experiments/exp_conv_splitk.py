"""Round 6 (third session): the stride-2 pyramid convs (256 -> 256, FD:150-162) at small batches - K slices as extra blocks
(ff3d_conv3x3_f16x3_splitk) against the one-pass kernels (the swapped-operand instance, default since this round; the 128 x 128 tiles with
FF3D_CONV_S2_SWAP=0 in the environment of the process).  Device time of 20 back-to-back launches inside one captured graph; error against
an fp64 convolution.
    python tools/experiments/exp_conv_splitk.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops  # noqa: E402

REP = 20


def graph_time(fn):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / (5 * REP)


print('FF3D_CONV_S2_SWAP =', os.environ.get('FF3D_CONV_S2_SWAP', '(default: swapped-operand instance)'))
gen = torch.Generator().manual_seed(0)
for (B, Cc, H, W, stride) in ((1, 256, 180, 180, 2), (1, 256, 90, 90, 2), (2, 256, 180, 180, 2), (2, 256, 90, 90, 2), (4, 256, 180, 180, 2),
                              (4, 256, 90, 90, 2), (8, 256, 90, 90, 2), (1, 128, 180, 180, 2), (1, 128, 90, 90, 2), (1, 256, 60, 60, 1)):
    x = torch.randn(B, Cc, H, W, generator=gen).cuda()
    w = (torch.randn(Cc, Cc, 3, 3, generator=gen) * 0.02).cuda()
    b = torch.randn(Cc, generator=gen).cuda()
    wp = ops.split_weight_f16(w, bias=b)
    xp = ops.split_f16(x, to_nhwc=True)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1))
    scale = float(ref.abs().max())
    line = 'B=%d C=%d %dx%d s%d:' % (B, Cc, H, W, stride)
    Ho = (H - 1) // stride + 1
    auto = ops.conv_ksplit(B * Ho * Ho, Cc, 9 * Cc)
    for ks in (1, 2, 3, 4, 6, 8, 12, 18):
        if ks > 9 * Cc // 32:
            continue
        os.environ['FF3D_CONV_KSPLIT_FORCE'] = str(ks)
        out = ops.conv3x3_f16x3(xp, wp, b, True, stride)
        err = float((out.double() - ref).abs().max()) / scale
        us = graph_time(lambda: ops.conv3x3_f16x3(xp, wp, b, True, stride))
        line += '  ks=%d%s %.1f us (%.1e)' % (ks, '*' if ks == auto else '', us, err)
    os.environ.pop('FF3D_CONV_KSPLIT_FORCE')
    print(line)
