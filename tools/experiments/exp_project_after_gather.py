"""Round 4 A/B (VERDICT r03 "next" #2): project-after-gather for the deformable cross-attention.

Today (per decoder stage of 3 layers, B = 32, C = 256, Nv = 42 525): bev_flatten writes the (raw + pos) pair, ONE weight-
stationary GEMM projects every cell for the three layers (M = B*Nv, K = 256, N = 768: 2 x 4.18 GB of fp32 stores per step),
and each layer's gather reads 4 corners x Dh = 32 channels of its head's slice per sample.
Alternative: value_proj is linear, so gather the UN-projected rows - every (query, head) pair gathers all C = 256 channels at
its own sampling locations - and project the gathered (B*Nq*heads, C) rows per head afterwards (0.5 GFLOP / frame).  The gather
then moves 8 x the bytes per sample (C instead of Dh channels per corner).

This script measures the gather side of the alternative with the SAME kernel (msda_fwd_kernel, one wave per pair: Dh = 256 ->
64 lanes x 16 B per corner row) by presenting every (query, head) pair as a one-head query over a (B, Nv, 1, 256) value tensor,
next to today's gather and today's GEMM + its share of the flatten, at B = 32 and B = 4.  Sampling locations: uniform
reference points + mmcv's ring-initialised offsets scaled like the bench's (queries of the synthetic workload are scattered).
Kill criterion (VERDICT): value path (flatten + GEMM + 6 gathers = 5.63 ms at B = 32 today) <= 3.6 ms."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops  # noqa: E402


def t(fn, n=10, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    res = []
    for B in (32, 4):
        Nq, M, L, P, C = 600, 8, 3, 4, 256
        hw = [(180, 180), (90, 90), (45, 45)]
        Nv = sum(h * w for h, w in hw)
        g = torch.Generator(device='cuda').manual_seed(B)
        ref = torch.rand(B, Nq, 1, 1, 1, 2, device='cuda', generator=g)
        off = torch.randn(B, Nq, M, L, P, 2, device='cuda', generator=g) * 0.02           # a few cells around the reference point
        loc = (ref + off).contiguous()
        w = torch.rand(B, Nq, M, L * P, device='cuda', generator=g).softmax(-1).view(B, Nq, M, L, P).contiguous()
        # today: projected value (B, Nv, 8, 32), one of the three column blocks of the stage GEMM's output (cell stride 768)
        wide = torch.randn(B, Nv, 3 * C, device='cuda', generator=g)
        v_today = wide[:, :, :C].unflatten(2, (M, C // M))
        out_today = torch.empty(B, Nq, C, device='cuda')
        v_dense = v_today.contiguous()
        ms_today = t(lambda: ops.msda_fwd(v_dense, hw, loc, w, out_today))
        # alternative: raw (+ pos) rows (B, Nv, 1, 256); (query, head) pairs as one-head queries
        raw = torch.randn(B, Nv, 1, C, device='cuda', generator=g)
        loc2 = loc.reshape(B, Nq * M, 1, L, P, 2).contiguous()
        w2 = w.reshape(B, Nq * M, 1, L, P).contiguous()
        out2 = torch.empty(B, Nq * M, C, device='cuda')
        ms_alt = t(lambda: ops.msda_fwd(raw, hw, loc2, w2, out2))
        # the per-head projection behind it: (B*Nq, heads, C) x (heads, C, Dh) -> (B*Nq, heads*Dh): vendor bmm as a stand-in
        wv = torch.randn(M, C, C // M, device='cuda', generator=g)
        g_rows = out2.view(B * Nq, M, C).transpose(0, 1)
        ms_proj = t(lambda: torch.bmm(g_rows, wv))
        alg_today = ops.msda_algorithmic_bytes(B, Nq, M, C // M, L, P, 4)
        alg_alt = ops.msda_algorithmic_bytes(B, Nq * M, 1, C, L, P, 4)
        res.append(dict(B=B, gather_today_ms=round(ms_today, 4), gather_unprojected_ms=round(ms_alt, 4),
                        per_head_projection_bmm_ms=round(ms_proj, 4), algorithmic_MB_today=round(alg_today / 1e6, 1),
                        algorithmic_MB_unprojected=round(alg_alt / 1e6, 1),
                        unprojected_algorithmic_TBps=round(alg_alt / ms_alt / 1e9, 2)))
        print(json.dumps(res[-1]), flush=True)
    b32 = res[0]
    today = 5.63                                  # flatten 1.24 + GEMM 2 x 1.83 + gather 6 x 0.12 (profiles/r03_o_*, B = 32)
    alt = 6 * (b32['gather_unprojected_ms'] + b32['per_head_projection_bmm_ms']) + 0.5    # + a flatten that writes raw rows only
    print(json.dumps(dict(value_path_today_ms=today, value_path_project_after_gather_ms=round(alt, 2),
                          kill_criterion_ms=3.6, verdict='keep today' if alt > 3.6 else 'build it')), flush=True)


if __name__ == '__main__':
    main()
