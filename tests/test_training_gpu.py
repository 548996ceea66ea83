"""Training targets and losses of the head on the HIP path (focalformer3d_amd/training.py) against
 (a) the golden produced by the REFERENCE's HungarianAssigner3D / get_targets / loss (tests/golden/train_targets.npz), and
 (b) the oracle's restatements for the two new kernels (3-D IoU, Gaussian heatmap targets) on seeded cases."""
import numpy as np
import pytest
import torch

from oracle import train_oracle as T
from tests.test_oracle_golden import load_train_golden
from tests.util import head_kwargs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from focalformer3d_amd import ops as o
    return o


def _boxes(g, n, spread=30.0):
    b = torch.zeros(n, 9)
    b[:, :2] = torch.rand(n, 2, generator=g) * spread - spread / 2
    b[:, 2] = torch.rand(n, generator=g) * 2 - 2
    b[:, 3:6] = torch.rand(n, 3, generator=g) * torch.tensor([2.0, 4.0, 1.5]) + torch.tensor([0.5, 0.8, 1.0])
    b[:, 6] = (torch.rand(n, generator=g) - 0.5) * 6.2
    return b


@pytest.mark.parametrize('n,m', [(200, 37), (1, 1), (60, 300)])
def test_boxes_iou3d_vs_oracle(ops, n, m):
    g = torch.Generator().manual_seed(n + m)
    a, b = _boxes(g, n), _boxes(g, m)[:, :7].contiguous()
    b[: min(m, 10)] = a[: min(m, 10), :7] + torch.randn(min(m, 10), 7, generator=g) * 0.05        # strongly overlapping pairs
    ref = T.boxes_iou3d(a, b)
    out = ops.boxes_iou3d(a.cuda(), b.cuda()).cpu()
    assert out.shape == ref.shape and float(ref.max()) > 0.3
    assert torch.allclose(out, ref, atol=2e-5, rtol=1e-4), (out - ref).abs().max()
    same = ops.boxes_iou3d(a.cuda(), a.cuda()).cpu()
    assert torch.allclose(same.diag(), torch.ones(n), atol=1e-4)


def test_gaussian_heatmap_targets_vs_oracle(ops):
    """FD:1133-1158 on the device vs the reference's host loop (restated): radius, truncated centres, clipped patches at the
    borders, overlapping boxes of one class, degenerate boxes skipped."""
    g = torch.Generator().manual_seed(4)
    K, H, W = 10, 180, 180
    tc = dict(grid_size=[1440, 1440, 40], voxel_size=[0.075, 0.075, 0.2], out_size_factor=8, gaussian_overlap=0.1,
              min_radius=2, point_cloud_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0])
    gt = _boxes(g, 80, spread=112.0)                     # some centres outside the range
    gt[:6, :2] = torch.tensor([[-54.0, -54.0], [53.99, 53.99], [-53.7, 10.0], [0.0, 53.9], [-60.0, 0.0], [20.0, 70.0]])
    gt[6, 3] = 0.0                                       # zero width: skipped by the reference
    gt[7:12, 3:5] = torch.tensor([12.0, 20.0])           # big boxes: large radii
    gt[12:16, :2] = gt[7:11, :2] + 0.3                   # overlapping gaussians of one class
    labels = torch.randint(0, K, (80,), generator=g)
    labels[12:16] = labels[7:11]
    ref = torch.zeros(K, H, W)
    vox, osf, pcr = torch.tensor(tc['voxel_size']), tc['out_size_factor'], torch.tensor(tc['point_cloud_range'])
    for i in range(len(gt)):
        width, length = gt[i][3] / vox[0] / osf, gt[i][4] / vox[1] / osf
        if width > 0 and length > 0:
            radius = max(tc['min_radius'], int(T.gaussian_radius((length, width), min_overlap=tc['gaussian_overlap'])))
            cx, cy = (gt[i][0] - pcr[0]) / vox[0] / osf, (gt[i][1] - pcr[1]) / vox[1] / osf
            T.draw_heatmap_gaussian(ref[labels[i]], torch.tensor([cx, cy], dtype=torch.float32).to(torch.int32), radius)
    out = ops.gaussian_heatmap_targets(gt.cuda(), labels.cuda(), K, H, W, (osf, 0.075, 0.075, -54.0, -54.0), 0.1, 2).cpu()
    assert float(ref.max()) == 1.0 and int((ref > 0).sum()) > 2000
    assert torch.equal(out, ref), (out - ref).abs().max()


@pytest.mark.parametrize('batched', [True, False])
def test_targets_and_losses_match_reference_golden(batched):
    """batched: head_get_targets_batched (two host round trips per batch) or the per-frame get_targets_single.
    FocalDecoder.get_targets / loss on the device: Hungarian assignment (cost matrix with the HIP IoU-3D kernel), box
    targets, heatmap targets, focal / L1 / Gaussian-focal losses - against what the reference's own code produced."""
    import focalformer3d_amd.focal_decoder  # noqa: F401
    from focalformer3d_amd.registry import build_head
    cfg, z, preds, gts, labels, ref, ref_losses = load_train_golden()
    h = dict(cfg['head'], input_img=False, iterbev_wo_img=True, multiscale=True, bevpos=True, mask_heatmap_mode='poscls')
    kw = head_kwargs(h)
    kw.update(train_cfg=cfg['train_cfg'], gt_center_limit=h['gt_center_limit'], add_gt_groups=0, **cfg['losses'])
    head = build_head(kw).cuda().eval()
    head.batched_targets = batched
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd/')}
    head.load_state_dict(sd, strict=False)
    dev = lambda t: [x.cuda() for x in t] if isinstance(t, list) else t.cuda()                        # noqa: E731
    p = {k: dev(v) for k, v in preds.items()}
    # as in the reference, the targets are defined relative to the last forward (FD:791 sets num_proposals = k * stages)
    inputs = [torch.from_numpy(z['in/pts_feat_conv']).cuda(), [torch.from_numpy(z[f'in/stage_{i}']).cuda() for i in range(3)]]
    mine = head(inputs, None, [{}, {}])
    assert head.num_proposals == 60
    got = head.get_targets([g.cuda() for g in gts], [l.cuda() for l in labels], [p])
    names = ('labels', 'label_weights', 'bbox_targets', 'bbox_weights', 'ious', 'num_pos', 'matched_ious', 'heatmap')
    for name, v in zip(names, got):
        r = ref[name]
        if torch.is_tensor(v):
            assert v.shape == r.shape, name
            if name == 'heatmap':
                assert torch.equal(v.cpu(), r), name
            else:
                assert torch.allclose(v.cpu().float(), r.float(), atol=2e-5, rtol=1e-4), name
        else:
            assert abs(float(v) - float(r)) < 1e-4, name
    losses = head.loss([g.cuda() for g in gts], [l.cuda() for l in labels], [[p]])
    assert set(losses) == set(ref_losses)
    for name, v in losses.items():
        assert abs(float(v) - ref_losses[name]) <= 2e-5 * max(1.0, abs(ref_losses[name])), (name, float(v), ref_losses[name])
    # the predictions of OUR forward give the same targets as the reference's predictions (forward parity is tested elsewhere)
    got2 = head.get_targets([g.cuda() for g in gts], [l.cuda() for l in labels], mine[0])
    assert int(got2[5]) == int(ref['num_pos'])
