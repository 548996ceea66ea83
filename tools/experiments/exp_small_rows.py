"""Round 6 (third session): device time of the decoder's dense steps at small row counts (one and four frames: 600 / 2 400 rows), each as
50 back-to-back launches inside one captured graph (no host in the timing): plain projection, projection + add + LayerNorm (K = 256 and
K = 1 024), the two-launch form of the same, the one-launch feed-forward.
    python tools/experiments/exp_small_rows.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops  # noqa: E402

REP = 50


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


gen = torch.Generator().manual_seed(0)
for M in (600, 1200, 2400, 4096):
    x = torch.randn(M, 256, generator=gen).cuda()
    x4 = torch.randn(M, 1024, generator=gen).cuda()
    res = torch.randn(M, 256, generator=gen).cuda()
    pos = torch.randn(M, 256, generator=gen).cuda()
    gam, bet = torch.ones(256).cuda(), torch.zeros(256).cuda()
    mk = lambda n, k: (torch.randn(n, k, generator=gen) * 0.05).cuda()
    w256, w768, w1024, wfc2 = mk(256, 256), mk(768, 256), mk(1024, 256), mk(256, 1024)
    b256, b768, b1024 = torch.randn(256, generator=gen).cuda(), torch.randn(768, generator=gen).cuda(), torch.randn(1024, generator=gen).cuda()
    s256, s768, s1024, sfc2 = (ops.split_weight_f16(w, bias=b) for w, b in ((w256, b256), (w768, b768), (w1024, b1024), (wfc2, b256)))
    wk4 = ops.kslice_weight(wfc2, 4)
    rows = [
        ('linear 256 -> 256', lambda: ops.linear_f16x3(x, s256, b256)),
        ('linear 256 -> 768 (q | k | v)', lambda: ops.linear_f16x3(x, s768, b768)),
        ('linear 256 -> 1024 + relu (fc1)', lambda: ops.linear_f16x3(x, s1024, b1024, True)),
        ('linear + add + LN, K = 256, one launch', lambda: ops.linear_add_ln_f16x3(x, s256, b256, res, gam, bet, pos=pos)),
        ('linear + add + LN, K = 256, two launches', lambda: ops.add_layer_norm(ops.linear_f16x3(x, s256, b256), res, gam, bet, pos=pos)),
        ('linear + add + LN, K = 1024 (fc2), one launch', lambda: ops.linear_add_ln_f16x3(x4, sfc2, b256, res, gam, bet, pos=pos)),
        ('linear + add + LN, K = 1024 (fc2), two launches', lambda: ops.add_layer_norm(ops.linear_f16x3(x4, sfc2, b256), res, gam, bet, pos=pos)),
        ('add + LN alone', lambda: ops.add_layer_norm(x, res, gam, bet, pos=pos)),
        ('linear + add + LN, K = 1024 (fc2), K slices: 4 column blocks + sum-LN', lambda: ops.sum_add_layer_norm(ops.linear_kslices_f16x3(x4, wk4, 4, 256), 4, b256, res, gam, bet, 1e-5, pos)),
    ]
    try:
        t1, t2 = ops.tile_weight_f16(w1024, bias=b1024), ops.tile_weight_f16(wfc2, bias=b256)
        rows.append(('feed-forward as ONE launch (ffn_rows)', lambda: ops.ffn_rows(x, t1, b1024, t2, b256, res, gam, bet, pos=pos)))
    except Exception as e:                                      # noqa: BLE001
        print('ffn_rows not set up here:', repr(e)[:200])
    for name, fn in rows:
        print('M = %4d  %-52s %7.2f us' % (M, name, graph_time(fn)))
