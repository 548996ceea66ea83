"""Training-mode forward / loss / backward of the head on the MI355X (focalformer3d_amd/train_forward.py): selection, sine
embeddings, deformable gather forward + backward, 3-D IoU and heatmap targets on libff3d_hip.so, the learnable layers on the
framework's autograd ops - against one training step executed by the REFERENCE (tests/golden/train_step_*.npz: predictions,
losses, BatchNorm buffers, the gradient of every parameter and input map)."""
import pytest
import torch

from tests.train_step_util import build_train_head, check_train_step, load_train_step, run_train_step

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,roi', [('train_step_nus', 'hip'), ('train_step_waymo', 'hip'), ('train_step_nus', 'grid_sample')])
def test_training_step_matches_reference(name, roi):
    """roi: the RoI feature read on ff3d_roi_grid_sample / _bwd (default) or on the framework's grid_sample (the reference's own
    op sequence) - both must reproduce the reference's step."""
    cfg, z = load_train_step(name)
    head = build_train_head(cfg)
    head.train_roi_sampler = roi
    p0, losses, grads, gin = run_train_step(head, z, 'cuda')
    if cfg['head'].get('add_gt_groups', 0):
        assert 'center_gtgroups' in p0 and p0['batch_valid_gt_mask'].dtype == torch.bool
    check_train_step(z, p0, losses, grads, gin, head, grad_atol_frac=5e-4)


def _full_size_head(train_cfg=True, **over):
    from focalformer3d_amd.synthetic import build_head_from_cfg, focalformer3d_l_head_cfg
    cfg = focalformer3d_l_head_cfg(C=128, grid=180, num_proposals=200, stages=3, decoder_stages=2)
    cfg.update(over)
    if train_cfg:
        cfg['train_cfg'] = dict(
            dataset='nuScenes',
            assigner=dict(type='HungarianAssigner3D', iou_calculator=dict(type='BboxOverlaps3D', coordinate='lidar'),
                          cls_cost=dict(type='FocalLossCost', gamma=2, alpha=0.25, weight=0.15),
                          reg_cost=dict(type='BBoxBEVL1Cost', weight=0.25), iou_cost=dict(type='IoU3DCost', weight=0.25)),
            pos_weight=-1, gaussian_overlap=0.1, min_radius=2, grid_size=[1440, 1440, 40], voxel_size=[0.075, 0.075, 0.2],
            out_size_factor=8, code_weights=[1.0] * 8 + [0.2, 0.2], point_cloud_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0])
    return build_head_from_cfg(cfg, seed=0, device='cuda'), cfg


@pytest.mark.parametrize('variant', ['nuscenes', 'waymo15_classaware'])
def test_train_route_equals_inference_route_at_full_size(variant):
    """With dropout off and BatchNorm on its running statistics the differentiable route (autograd ops + HIP gather) and the
    inference route (hand-written kernels end to end) are two implementations of one function: FocalFormer3D-L shape, C = 128;
    also with the class-aware regression heads of FocalFormer3D_Waymo15_L (K = 3, no velocity)."""
    from focalformer3d_amd.synthetic import build_head_from_cfg, focalformer3d_l_head_cfg, stage_features
    if variant == 'nuscenes':
        head, _ = _full_size_head(train_cfg=False, add_gt_groups=0)
    else:
        cfg = focalformer3d_l_head_cfg(C=128, grid=180, num_proposals=200, stages=3, decoder_stages=2, num_classes=3,
                                       dataset='Waymo')
        cfg.update(classaware_reg=True, add_gt_groups=0)
        head = build_head_from_cfg(cfg, seed=0, device='cuda')
    inputs = stage_features(2, 128, 180, 3, seed=1, device='cuda')
    ev = head.eval()(inputs, None, [{}] * 2)[0][0]
    ev_labels = head.query_labels.clone()
    head.train()
    for m in head.modules():          # batch statistics and every dropout off; the wrappers stay on their training route
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.Dropout, torch.nn.MultiheadAttention)):
            m.eval()
    with torch.no_grad():
        tr = head(inputs, None, [{}] * 2)[0][0]
    assert set(tr) == set(ev)
    # the two routes compute the heatmap logits with different kernels (MIOpen fp32 vs split-fp16): candidates whose scores agree
    # to round-off may swap ranks inside a stage's top-k - compare under that permutation (tests/util.align_queries)
    from tests.util import align_queries, permute_queries
    nq = head.query_labels.shape[1]
    tr_labels = head.query_labels.clone()
    perm = align_queries({'center': tr['center'].detach()}, {'center': ev['center']}, tr_labels, ev_labels, nq, nq // 3, max_moved=12)
    assert torch.equal(tr_labels.cpu(), permute_queries(ev_labels.cpu(), perm, nq))
    for key in ('center', 'height', 'dim', 'rot', 'vel', 'heatmap', 'query_heatmap_score'):
        if key not in ev:
            continue
        scale = max(1.0, float(ev[key].abs().max()))
        assert float((tr[key].cpu() - permute_queries(ev[key].cpu(), perm, nq)).abs().max()) <= 2e-4 * scale, key
    for a, b in zip(tr['dense_heatmap'], ev['dense_heatmap']):
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max()))
    for a, b in zip(tr['multistage_masks'], ev['multistage_masks']):
        assert torch.equal(a, b)


def test_full_size_training_step_runs_and_updates():
    """FocalFormer3D-L head (C = 128, 600 queries + 3 ground-truth groups, dropout 0.1 as configured): forward with ground truth,
    loss, backward, one AdamW step - finite losses, a gradient on every parameter, the parameters move, BatchNorm buffers move,
    and a second step on the updated weights (rebuilt weight caches) works."""
    from focalformer3d_amd.synthetic import stage_features
    head, cfg = _full_size_head()
    head.train()
    B = 2
    inputs = stage_features(B, 128, 180, 3, seed=2, device='cuda')
    g = torch.Generator().manual_seed(3)
    gts, labels = [], []
    for b in range(B):
        n = 20 + 11 * b
        t = torch.zeros(n, 9)
        t[:, :2] = torch.rand(n, 2, generator=g) * 100 - 50
        t[:, 2] = torch.rand(n, generator=g) * 2 - 2.5
        t[:, 3:6] = torch.rand(n, 3, generator=g) * torch.tensor([2.0, 4.0, 1.5]) + torch.tensor([0.6, 0.8, 1.0])
        t[:, 6] = (torch.rand(n, generator=g) - 0.5) * 6.2
        gts.append(t.cuda())
        labels.append(torch.randint(0, 10, (n,), generator=g).cuda())
    opt = torch.optim.AdamW(head.parameters(), lr=1e-4, weight_decay=0.01)
    before = {n: p.detach().clone() for n, p in head.named_parameters()}
    bn_before = head.heatmap_head[0].bn.running_mean.clone()
    for step in range(2):
        opt.zero_grad()
        preds = head(inputs, None, [{}] * B, gt_bboxes_3d=gts, gt_labels_3d=labels)
        p = preds[0][0]
        assert p['center'].shape == (B, 2, 2 * 600) and p['center_gtgroups'].shape == (B, 2, 2 * 3 * 31)
        losses = head.loss(gts, labels, preds)
        total = sum(v for n, v in losses.items() if 'loss' in n)
        assert torch.isfinite(total)
        assert {'loss_heatmap', 'layer_0_loss_cls', 'layer_1_loss_bbox', 'gt_query_loss_box', 'gt_query_loss_cls'} <= set(losses)
        total.backward()
        missing = [n for n, q in head.named_parameters() if q.grad is None]
        assert not missing, missing
        assert all(bool(torch.isfinite(q.grad).all()) for q in head.parameters())
        torch.nn.utils.clip_grad_norm_(head.parameters(), 35.0)
        opt.step()
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in head.named_parameters())
    assert moved == len(before)
    assert not torch.equal(bn_before, head.heatmap_head[0].bn.running_mean)
    # the inference path sees the updated weights (weight-signature caches rebuilt) and still works
    out = head.eval()(inputs, None, [{}] * B)
    assert bool(torch.isfinite(out[0][0]['center']).all())
