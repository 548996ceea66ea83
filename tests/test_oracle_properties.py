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


def test_rotated_iou_matches_independent_polygon_clipping():
    """The restated mmdet3d box_overlap (edge intersections + corner containment + angular sort, fp32) against an
    independent fp64 Sutherland-Hodgman clip of the same rectangles; plus symmetry and the self-IoU."""
    def corners64(b):
        cx, cy = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2
        pts = np.array([[b[0], b[1]], [b[2], b[1]], [b[2], b[3]], [b[0], b[3]]], dtype=np.float64)
        c, s = np.cos(b[4]), np.sin(b[4])
        d = pts - [cx, cy]
        return np.stack([d[:, 0] * c + d[:, 1] * s + cx, -d[:, 0] * s + d[:, 1] * c + cy], 1)

    def shoelace(p):
        return 0.5 * np.sum(p[:, 0] * np.roll(p[:, 1], -1) - np.roll(p[:, 0], -1) * p[:, 1])

    def clip_area(subject, clipper):
        if shoelace(clipper) < 0:
            clipper = clipper[::-1]
        out = [tuple(p) for p in subject]
        for i in range(4):
            a, b = clipper[i], clipper[(i + 1) % 4]
            inp, out = out, []
            if not inp:
                break
            side = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])   # noqa: E731
            S = inp[-1]
            for E in inp:
                if (side(E) >= 0) != (side(S) >= 0):
                    t = side(S) / (side(S) - side(E))
                    out.append((S[0] + t * (E[0] - S[0]), S[1] + t * (E[1] - S[1])))
                if side(E) >= 0:
                    out.append(E)
                S = E
        return abs(shoelace(np.array(out))) if len(out) >= 3 else 0.0

    rng = np.random.default_rng(3)
    n = 40
    c, wh, r = rng.uniform(-5, 5, (n, 2)), rng.uniform(0.5, 5, (n, 2)), rng.uniform(-4, 4, n)
    b = np.concatenate([c - wh / 2, c + wh / 2, r[:, None]], 1).astype(np.float32)
    iou = O.boxes_iou_bev(b, b)
    ref = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            ov = clip_area(corners64(b[i].astype(np.float64)), corners64(b[j].astype(np.float64)))
            sa, sb = (b[i, 2] - b[i, 0]) * (b[i, 3] - b[i, 1]), (b[j, 2] - b[j, 0]) * (b[j, 3] - b[j, 1])
            ref[i, j] = ov / max(sa + sb - ov, 1e-8)
    assert np.abs(iou - ref).max() < 5e-6
    assert np.abs(iou - iou.T).max() < 5e-6 and np.abs(iou.diagonal() - 1).max() < 5e-6
    assert (ref > 0).mean() > 0.1


def test_rotated_nms_properties():
    """nms_bev: kept boxes are mutually below the threshold, every dropped box overlaps a better kept one, order = score."""
    rng = np.random.default_rng(4)
    n = 300
    c, wh, r = rng.uniform(0, 12, (n, 2)), rng.uniform(0.5, 3, (n, 2)), rng.uniform(-4, 4, n)
    b = np.concatenate([c - wh / 2, c + wh / 2, r[:, None]], 1).astype(np.float32)
    s = rng.uniform(0, 1, n).astype(np.float32)
    iou = O.boxes_iou_bev(b, b)
    keep = O.nms_bev(b, s, 0.2)
    k = np.array(keep)
    assert (np.diff(s[k]) <= 0).all()
    sub = iou[np.ix_(k, k)].copy()
    np.fill_diagonal(sub, 0)
    assert (sub <= 0.2).all()
    dropped = np.setdiff1d(np.arange(n), k)
    for d in dropped:
        better = k[s[k] >= s[d]]
        assert (iou[d, better] > 0.2).any()
    order = np.argsort(-s, kind='stable')[:50]                      # pre / post caps = NMS of the 50 best, first 7 kept
    sub_keep = O.nms_bev(b[order], s[order], 0.2)
    assert O.nms_bev(b, s, 0.2, pre_maxsize=50, post_max_size=7) == [int(order[i]) for i in sub_keep][:7]


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 2), st.integers(1, 6), st.integers(0, 2 ** 31 - 1))
def test_points_in_boxes_symmetries_and_independent_polygon_test(B, T, seed):
    """The restated mmdet3d points_in_boxes_gpu (rotate by rz + pi / 2, local_x against l, local_y against w) against an independent
    fp64 half-plane test of the rectangle's corners; turning a box by pi, or by pi / 2 with w <-> l swapped, leaves the footprint
    unchanged; translating boxes and points together changes nothing (points kept 1e-3 away from every edge)."""
    rng = np.random.default_rng(seed)
    boxes = np.zeros((B, T, 7), np.float32)
    boxes[..., :2] = rng.uniform(-10, 10, (B, T, 2))
    boxes[..., 2] = -1.0
    boxes[..., 3:5] = rng.uniform(0.7, 8, (B, T, 2))
    boxes[..., 5] = 2.0
    boxes[..., 6] = rng.uniform(-3.1, 3.1, (B, T))
    pts = np.zeros((B, 300, 3), np.float32)
    pts[..., :2] = rng.uniform(-14, 14, (B, 300, 2))

    def independent(p, b):
        out = np.full(p.shape[:2], -1, np.int64)
        near = np.zeros(p.shape[:2], bool)
        for bi in range(p.shape[0]):
            for t in reversed(range(b.shape[1])):
                cx, cy, _, w, l, _, rz = b[bi, t].astype(np.float64)
                a = rz + np.pi / 2                                  # the box's long axis (l) points along (cos a, -sin a)
                ul, uw = np.array([np.cos(a), -np.sin(a)]), np.array([np.sin(a), np.cos(a)])
                d = p[bi, :, :2].astype(np.float64) - [cx, cy]
                dl, dw = np.abs(d @ ul) - l / 2, np.abs(d @ uw) - w / 2
                out[bi, (dl < 0) & (dw < 0)] = t                     # (reversed loop: the first box wins)
                near[bi] |= (np.abs(dl) < 1e-3) & (dw < 1e-3) | (np.abs(dw) < 1e-3) & (dl < 1e-3)
        return out, near
    want, near = independent(pts, boxes)
    tb, tp = torch.from_numpy(boxes), torch.from_numpy(pts)
    got = O.points_in_boxes(tp, tb).numpy()
    ok = ~near
    assert (got[ok] == want[ok]).all()
    assert (want >= 0).any() or T == 1
    turned = boxes.copy()
    turned[..., 6] += np.float32(np.pi)
    assert (O.points_in_boxes(tp, torch.from_numpy(turned)).numpy()[ok] == want[ok]).all()
    swapped = boxes.copy()
    swapped[..., 3], swapped[..., 4] = boxes[..., 4], boxes[..., 3]
    swapped[..., 6] += np.float32(np.pi / 2)
    assert (O.points_in_boxes(tp, torch.from_numpy(swapped)).numpy()[ok] == want[ok]).all()
    shift = np.array([3.25, -1.5, 0], np.float32)
    moved_b, moved_p = boxes.copy(), pts + shift
    moved_b[..., :3] += shift
    assert (O.points_in_boxes(torch.from_numpy(moved_p), torch.from_numpy(moved_b)).numpy()[ok] == want[ok]).all()


@settings(max_examples=15, deadline=None)
@given(st.integers(1, 2), st.integers(1, 8), st.integers(8, 20), st.integers(0, 2 ** 31 - 1))
def test_boxcls_mask_only_clears_contains_poscls_and_is_idempotent(B, k, H, seed):
    """FD:732-782: the 'boxcls' update clears a superset of what 'poscls' clears, never sets a cell, and applying it twice changes
    nothing; a query whose box (after the margin) still spans 2 x 2 m blanks at least its own cell's neighbours of its class."""
    g = torch.Generator().manual_seed(seed)
    K, W, HW = 10, H, H * H
    vox = 108.0 / (H * 8)
    cfg = O.head_config(num_classes=K, dataset='nuScenes', pc_range=(-54.0, -54.0), voxel_size=(vox, vox), out_size_factor=8)
    acc = (torch.rand(B, K * HW, generator=g) > 0.3).float()
    idx = torch.stack([torch.randperm(K * HW, generator=g)[:k] for _ in range(B)])
    cls, cell = idx // HW, idx % HW
    bev_pos = O.create_2d_grid(H, W).repeat(B, 1, 1)
    qb = torch.zeros(B, 10, k)
    qb[:, 0] = (cell % W).float() + torch.rand(B, k, generator=g)
    qb[:, 1] = (cell // W).float() + torch.rand(B, k, generator=g)
    qb[:, 3:6] = torch.rand(B, 3, k, generator=g) * 2.5
    ang = torch.rand(B, k, generator=g) * 6.28
    qb[:, 6], qb[:, 7] = torch.sin(ang), torch.cos(ang)
    small = O.SMALL_CLASSES['nuScenes']
    sel = O.box_class_mask(qb, cls, bev_pos, K, cfg)
    assert ((sel == 0) | (sel == 1)).all() and (sel.view(B, K, HW).sum(1) <= 1).all()        # a cell takes ONE query's class
    new = O.mask_update(acc, idx, K, H, W, 'boxcls', 3, small, box_sel=sel)
    pos = O.mask_update(acc, idx, K, H, W, 'poscls', 3, small)
    assert ((new == 0) | (new == acc)).all() and (new <= pos).all()
    assert torch.equal(O.mask_update(new, idx, K, H, W, 'boxcls', 3, small, box_sel=sel), new)
