"""Property tests (hypothesis) of the oracle's Hard-Instance-Probing invariants - the same invariants the HIP kernels
are checked against at full size where an element-wise CPU comparison would be too slow."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from oracle import ff3d_oracle as O


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 3), st.integers(1, 10), st.integers(3, 12), st.integers(3, 12), st.integers(0, 2 ** 31 - 1))
def test_nms_invariants(B, K, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    heat = torch.rand(B, K, H, W, generator=g)
    small = [c for c in (8, 9) if c < K]
    out = O.local_max_nms(heat, 3, small)
    assert ((out == 0) | (out == heat)).all()                       # survivors keep their score
    for c in range(K):
        if c in small:
            assert torch.equal(out[:, c], heat[:, c])                # kernel-1 classes pass through
        else:
            assert (out[:, c, 0] == 0).all() and (out[:, c, -1] == 0).all()      # border ring suppressed (FD:673-676)
            assert (out[:, c, :, 0] == 0).all() and (out[:, c, :, -1] == 0).all()
    inner = out[:, :, 1:-1, 1:-1]
    pooled = torch.nn.functional.max_pool2d(heat, 3, 1, 0)
    nsm = [c for c in range(K) if c not in small]
    assert ((inner[:, nsm] == 0) | (inner[:, nsm] == pooled[:, nsm])).all()      # survivors are 3x3 maxima


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 3), st.integers(5, 400), st.integers(0, 2 ** 31 - 1), st.floats(0.0, 0.95))
def test_topk_is_maximal_and_deterministic(B, n, seed, zero_frac):
    g = torch.Generator().manual_seed(seed)
    h = torch.rand(B, n, generator=g)
    h[torch.rand(B, n, generator=g) < zero_frac] = 0.0
    h = (h * 8).round() / 8                                          # many exact ties
    k = max(1, n // 3)
    idx = O.topk_deterministic(h, k)
    for b in range(B):
        sel = set(idx[b].tolist())
        assert len(sel) == k
        thr = min(h[b, i].item() for i in sel)
        rest = [i for i in range(n) if i not in sel]
        assert all(h[b, i] <= thr for i in rest)                     # nothing better left outside
        tied_out = [i for i in rest if h[b, i] == thr]
        tied_in = [i for i in sel if h[b, i] == thr]
        if tied_out:
            assert max(tied_in) < min(tied_out)                      # ties resolved to the lowest indices
        v = h[b, idx[b]]
        assert (v[:-1] >= v[1:]).all()                               # emitted in descending score order


@settings(max_examples=20, deadline=None)
@given(st.integers(1, 2), st.integers(2, 10), st.integers(4, 10), st.integers(0, 2 ** 31 - 1), st.sampled_from(['poscls', 'pos']))
def test_mask_update_only_clears_and_is_idempotent(B, K, H, seed, mode):
    g = torch.Generator().manual_seed(seed)
    W = H + 1
    acc = (torch.rand(B, K * H * W, generator=g) > 0.3).float()
    idx = torch.stack([torch.randperm(K * H * W, generator=g)[:5] for _ in range(B)])
    small = [c for c in (8, 9) if c < K]
    new = O.mask_update(acc, idx, K, H, W, mode, 3, small)
    assert ((new == 0) | (new == acc)).all() and (new <= acc).all()  # masks only ever lose cells
    assert torch.equal(O.mask_update(new, idx, K, H, W, mode, 3, small), new)
    for b in range(B):
        for f in idx[b].tolist():
            assert new[b, f] == 0                                     # every selected cell is masked out
