"""GPU parity of the heatmap_box branch (heatmap_box + thin_heatmap_box, mask_heatmap_mode='boxcls'; reference FD:231-287, 606-660,
708-722, 732-770): the two entry points through the C ABI against the oracle at the reference's sizes, and the whole head against
 (a) the goldens the REFERENCE module produced (tests/golden/head_opt_heatbox.npz, head_opt_boxcls.npz) and
 (b) the oracle at a second size and seed."""
import pytest
import torch

from oracle import ff3d_oracle as O
from tests.util import Boxes, dense_pairs, head_inputs, head_kwargs, load_golden, oracle_cfg, stage_perm

pytestmark = pytest.mark.gpu


def _ops():
    from focalformer3d_amd import ops
    return ops


def _cfg(H, dataset='nuScenes'):
    vox = 108.0 / (H * 8)
    return O.head_config(num_classes=10, dataset=dataset, pc_range=(-54.0, -54.0), voxel_size=(vox, vox), out_size_factor=8,
                         heatmap_box=True, thin_heatmap_box=True, mask_heatmap_mode='boxcls')


@pytest.mark.parametrize('H,k,B', [(180, 200, 4), (24, 12, 2), (37, 5, 1)])
def test_heatmap_box_gather_matches_oracle(H, k, B):
    """FD:708-722 bit for bit: task -> class expansion, cell offsets, the four clips (values drawn wide enough to hit every clip),
    gather at arbitrary (class, cell) proposals; columns outside [q_offset, q_offset + k) stay untouched."""
    ops = _ops()
    g = torch.Generator().manual_seed(H + k)
    K, W = 10, H
    raw = torch.randn(B, 60, H, W, generator=g) * 6.0
    idx = torch.stack([torch.randperm(K * H * W, generator=g)[:k] for _ in range(B)])
    ref = O.heatmap_box_gather(raw, idx, O.create_2d_grid(H, W).repeat(B, 1, 1), K)
    Nq, q0 = 3 * k, k
    out = torch.full((B, 10, Nq), -777.0).cuda()
    ops.heatmap_box_gather(raw.cuda(), idx.cuda(), out, q0, K)
    assert torch.equal(out[:, :, q0:q0 + k].cpu(), ref)
    assert (out[:, :, :q0] == -777.0).all() and (out[:, :, q0 + k:] == -777.0).all()
    assert (ref[:, 2].abs() <= 5).all() and (ref[:, 3:6] == ref[:, 3:6].max()).any() and (ref[:, 6:8].abs() <= 1).all()


def _random_boxes(B, k, H, g):
    """Query boxes in head coordinates: centres on the grid, log-dims up to the clip, arbitrary headings."""
    qb = torch.zeros(B, 10, k)
    qb[:, 0:2] = torch.rand(B, 2, k, generator=g) * H
    qb[:, 2] = torch.randn(B, k, generator=g)
    qb[:, 3:6] = torch.rand(B, 3, k, generator=g) * 2.9 - 0.2
    ang = torch.rand(B, k, generator=g) * 6.283 - 3.1415
    qb[:, 6], qb[:, 7] = torch.sin(ang), torch.cos(ang)
    qb[:, 8:10] = torch.randn(B, 2, k, generator=g)
    return qb


@pytest.mark.parametrize('H,k,B,ks', [(180, 200, 4, 3), (48, 40, 2, 3), (48, 40, 2, 1)])
def test_box_class_mask_matches_oracle(H, k, B, ks):
    """FD:732-768 + FD:774-782: the in-place clear equals acc * (1 - maxpool(box mask)) of the oracle (points_in_boxes restatement),
    bit for bit except cells whose centre lies within 1e-3 m of a box edge (float32 round-off of exp / atan2 / cos / sin differs between
    the CPU and the GPU there; the reference's own CUDA kernel has the same freedom) - counted, and at most a handful."""
    ops = _ops()
    g = torch.Generator().manual_seed(7 * H + k)
    K, W, HW = 10, H, H * H
    cfg = _cfg(H)
    small = O.SMALL_CLASSES['nuScenes']
    qb = _random_boxes(B, k, H, g)
    labels = torch.randint(0, K, (B, k), generator=g)
    acc = (torch.rand(B, K * HW, generator=g) > 0.2).float()
    bev_pos = O.create_2d_grid(H, W).repeat(B, 1, 1)
    sel = O.box_class_mask(qb, labels, bev_pos, K, cfg)
    assert 0.001 < sel.mean() < 0.5, 'the case must mask a visible share of the cells'
    none = torch.zeros(B, 0, dtype=torch.int64)
    ref = O.mask_update(acc, none, K, H, W, 'boxcls', ks, small, box_sel=sel).view(B, K, H, W)
    Nq, q0 = 3 * k, 2 * k
    qb_all = torch.zeros(B, 10, Nq)
    qb_all[:, :, q0:q0 + k] = qb
    lab_all = torch.zeros(B, Nq, dtype=torch.int64)
    lab_all[:, q0:q0 + k] = labels
    mask = acc.view(B, K, H, W).clone().cuda()
    coder = (8.0, cfg.voxel_size[0], cfg.voxel_size[1], -54.0, -54.0)
    ops.box_class_mask(qb_all.cuda(), lab_all.cuda(), mask, k, q0, coder, (-54.0, -54.0, 54.0, 54.0), ks, ops.small_class_bits('nuScenes', K))
    diff = int((mask.cpu() != ref).sum())
    assert diff <= 9 * 4, f'{diff} mask cells differ'          # (a flipped edge cell moves at most its 3 x 3 window)
    if diff:
        # every difference must trace back to an edge case: recompute the oracle in float64 and require a centre within 1e-3 m of an edge
        import numpy as np
        std = O.decode_box(qb[:, 6:8].clone(), qb[:, 3:6].clone(), qb[:, 0:2].clone(), qb[:, 2:3].clone(), None, cfg).double()
        cx, cy = std[..., 0].clip(-54, 54), std[..., 1].clip(-54, 54)
        w, l = (std[..., 3] - 1).clip(0.7, 10), (std[..., 4] - 1).clip(0.7, 10)
        rot = std[..., 6] + np.pi / 2
        px = (bev_pos[..., 0].double() * 8 * cfg.voxel_size[0] - 54)[:, :, None]
        py = (bev_pos[..., 1].double() * 8 * cfg.voxel_size[1] - 54)[:, :, None]
        sx, sy = px - cx[:, None], py - cy[:, None]
        lx = sx * rot.cos()[:, None] - sy * rot.sin()[:, None]
        ly = sx * rot.sin()[:, None] + sy * rot.cos()[:, None]
        edge = torch.minimum((lx.abs() - l[:, None] / 2).abs(), (ly.abs() - w[:, None] / 2).abs())
        near = (edge < 1e-3) & (lx.abs() < l[:, None] / 2 + 1e-3) & (ly.abs() < w[:, None] / 2 + 1e-3)
        assert near.any(), 'mask cells differ without any cell centre near a box edge'


def _build(cfg, sd):
    import focalformer3d_amd.focal_decoder  # noqa: F401
    from focalformer3d_amd.registry import build_head
    head = build_head(head_kwargs(cfg))
    missing, unexpected = head.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    return head.cuda().eval()


def _cuda(inputs):
    return [inputs[0].cuda(), [t.cuda() for t in inputs[1]]]


@pytest.mark.parametrize('name', ['head_opt_heatbox', 'head_opt_boxcls'])
def test_head_matches_reference_golden(name):
    """The head with the heatmap_box branch on against what the REFERENCE module returned: bit-exact indices / labels / masks (the
    'boxcls' fixture's cell centres keep >= 3.8 mm from every box edge), predictions within north_star's 1e-4, and the branch's own
    outputs (FD:988-991): the per-stage / per-task dense box predictions, query_pos, query_box."""
    cfg, sd, inp, ref, _ = load_golden(name)
    head = _build(cfg, sd)
    out = head(_cuda(head_inputs(cfg, inp)), None, [{}] * 2)[0][0]
    k, K, H = cfg['num_proposals'], cfg['num_classes'], cfg['grid']
    n_st = int(head.multistage_heatmap)
    nq = k * n_st
    labels = head.query_labels.cpu()
    taps = {}
    with torch.no_grad():
        O.focal_decoder_forward(sd, oracle_cfg(cfg), head_inputs(cfg, inp), taps)
    perms = []
    for i in range(n_st):
        st = taps['stages'][i]
        v = torch.sort(st['heat'].reshape(2, -1), descending=True).values
        assert ((v[:, k - 1] - v[:, k]) > 1e-6).all()
        perms.append(stage_perm(ref[f'topk/{i}'][:, :k], st['idx']) + i * k)
        assert torch.equal(labels[:, i * k:(i + 1) * k], st['idx'] // (H * H)), 'query labels must be bit-exact'
    perm = torch.cat(perms, 1)
    assert torch.equal(ref['query_labels'].gather(1, perm), labels)
    for i, m in enumerate(out['multistage_masks']):
        assert torch.equal(m.cpu().to(torch.uint8), ref[f'multistage_masks/{i}']), 'masks must be bit-exact'
    if name == 'head_opt_boxcls':                                # the box part must matter: more blanked cells than the 3 x 3 windows alone
        assert int((ref['multistage_masks/2'] == 0).sum()) > 2 * 2 * k * 9
    D = cfg['num_decoder_layers']
    full = torch.cat([perm + d * nq for d in range(D)], 1)
    for key in list(cfg['common_heads'].keys()) + ['heatmap']:
        r = ref[key].gather(2, full[:, None, :].expand(-1, ref[key].shape[1], -1))
        assert torch.allclose(out[key].cpu(), r, atol=1e-4, rtol=1e-4), key
    for h, r in dense_pairs(out, ref):
        assert torch.allclose(h.cpu(), r, atol=1e-4, rtol=1e-4)
    assert len(out['multistage_bev_preds']) == n_st
    for i, tasks in enumerate(out['multistage_bev_preds']):
        assert len(tasks) == 6 and list(tasks[0].keys()) == ['reg', 'height', 'dim', 'rot', 'vel']
        got = torch.cat([torch.cat(list(t.values()), 1) for t in tasks], 1)
        assert torch.allclose(got.cpu(), ref[f'multistage_bev_preds/{i}'], atol=1e-4, rtol=1e-4)
    assert torch.allclose(out['query_box'].cpu(), ref['query_box'].gather(2, perm[:, None, :].expand(-1, 10, -1)), atol=1e-4, rtol=1e-4)
    assert torch.allclose(out['query_pos'].cpu(), ref['query_pos'].gather(1, perm[:, :, None].expand(-1, -1, 2)), atol=1e-4, rtol=1e-4)
    # get_bboxes of frame 0 (the reference asserts batch 1)
    fullin = head_inputs(cfg, inp)
    preds = head(_cuda([fullin[0][:1], [t[:1] for t in fullin[1]]]), None, [{}])
    (boxes, scores, lab), = head.get_bboxes(preds, [{'box_type_3d': Boxes}])
    import numpy as np
    b, rb = boxes.tensor.cpu(), ref['bboxes0']
    a, c = np.lexsort((b[:, 1].numpy(), b[:, 0].numpy())), np.lexsort((rb[:, 1].numpy(), rb[:, 0].numpy()))
    assert b.shape == rb.shape and torch.allclose(b[a], rb[c], atol=1e-4, rtol=1e-4)
    assert torch.allclose(scores.cpu()[a], ref['scores0'][c], atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize('mode,C,dense', [('boxcls', 32, 'f16x3'), ('poscls', 32, 'f16x3'), ('boxcls', 16, 'vendor')])
def test_head_heatmap_box_vs_oracle_second_size(mode, C, dense):
    """A second size and seed against the oracle: C = 32 runs the task heads on the split-fp16 kernels (wide conv -> pair -> four
    15-channel tail launches), 48 x 48 cells of 2.25 m so that the boxes cover several cells."""
    from focalformer3d_amd.synthetic import build_head_from_cfg, focalformer3d_l_head_cfg, stage_features
    from tests.util import oracle_cfg_from_head_cfg
    k, H = 16, 48
    hc = focalformer3d_l_head_cfg(C=C, grid=H, num_proposals=k, stages=3, decoder_stages=2, ffn=64, hidden_channel_roi=48)
    vox = 108.0 / (H * 8)
    hc['bbox_coder'].update(voxel_size=[vox, vox], pc_range=[-54.0, -54.0])
    hc['test_cfg'].update(voxel_size=[vox, vox], pc_range=[-54.0, -54.0])
    hc.update(heatmap_box=True, thin_heatmap_box=True, mask_heatmap_mode=mode)
    head = build_head_from_cfg(hc, seed=21)
    with torch.no_grad():                                   # wider task-head outputs: dims up to the clip, boxes over several cells
        for m in head.multi_stage_task_heads:
            m[1].weight.mul_(2.5)
            m[1].bias.add_(0.5)
    head.invalidate_cache()
    if dense == 'vendor':
        head.set_dense_mode('vendor')
    sd = {n: v.clone() for n, v in head.state_dict().items()}
    inputs = stage_features(2, C, H, 3, seed=22)
    ocfg = oracle_cfg_from_head_cfg(hc)
    ocfg.heatmap_box = ocfg.thin_heatmap_box = True
    taps = {}
    with torch.no_grad():
        ref, aux = O.focal_decoder_forward(sd, ocfg, inputs, taps)
    for st in taps['stages']:
        v = torch.sort(st['heat'].reshape(2, -1), descending=True).values
        assert ((v[:, k - 1] - v[:, k]) > 1e-6).all()
    head = head.cuda()
    out = head(_cuda(inputs), None, [{}] * 2)[0][0]
    assert torch.equal(head.query_labels.cpu(), aux['query_labels'])
    for m, r in zip(out['multistage_masks'], ref['multistage_masks']):
        assert torch.equal(m.cpu(), r)
    if mode == 'boxcls':                                    # the box part must matter: the masks differ from the 'poscls' ones
        ocfg.mask_heatmap_mode = 'poscls'
        with torch.no_grad():
            ref_pos, _ = O.focal_decoder_forward(sd, ocfg, inputs)
        assert sum(int((a != b).sum()) for a, b in zip(ref_pos['multistage_masks'], ref['multistage_masks'])) > 20
    for key in ('center', 'height', 'dim', 'rot', 'vel', 'heatmap', 'query_heatmap_score', 'query_box', 'query_pos'):
        assert out[key].shape == ref[key].shape, key
        assert torch.allclose(out[key].cpu(), ref[key], atol=1e-4, rtol=1e-4), key
    for tasks, r in zip(out['multistage_bev_preds'], ref['multistage_bev_preds']):
        got = torch.cat([torch.cat(list(t.values()), 1) for t in tasks], 1)
        assert torch.allclose(got.cpu(), r, atol=1e-4, rtol=1e-4)


def test_heatmap_box_head_as_one_graph_equals_eager():
    """The branch inside a captured head (runtime.GraphedHead): the two new launches take host tables by value, no allocation
    or synchronisation of theirs - replay == eager, bit for bit."""
    from focalformer3d_amd.runtime import GraphedHead
    cfg, sd, inp, ref, _ = load_golden('head_opt_boxcls')
    head = _build(cfg, sd)
    inputs = _cuda(head_inputs(cfg, inp))
    eager = head(inputs, None, [{}] * 2)
    eb = head.get_bboxes_padded(eager)
    gh = GraphedHead(head, inputs)
    res = gh()
    torch.cuda.synchronize()
    for a, b in zip(res, eb):
        assert torch.equal(a, b)


def test_head_heatmap_box_full_size_vs_oracle():
    """The branch at the reference's real sizes - 180 x 180 BEV (0.6 m cells), 3 stages x 200 queries, C = 128, 'boxcls' - against the
    oracle: labels and masks bit-exact, the box part blanking thousands of cells the 'poscls' rule leaves, predictions within 1e-4."""
    from tests.test_head_gpu import _full_size_case
    from tests.util import align_queries, permute_queries
    import focalformer3d_amd.focal_decoder  # noqa: F401
    from focalformer3d_amd.registry import build_head
    C, k, nq = 128, 200, 600
    cfg, _, _, inputs = _full_size_case(C, B=1, seed=5)
    cfg = dict(cfg, heatmap_box=True, thin_heatmap_box=True, mask_heatmap_mode='boxcls')
    torch.manual_seed(5)
    head = build_head(head_kwargs(cfg)).eval()
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for n, p in head.named_parameters():
            if p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) * (0.7 / max(1, p[0].numel()) ** 0.5))
            elif n.endswith('weight'):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        for n, b in head.named_buffers():
            if n.endswith('running_mean'):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            if n.endswith('running_var'):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
        for m in head.multi_stage_task_heads:               # boxes of a few metres: log-dims around 1
            m[1].weight.mul_(1.5)
            m[1].bias.add_(1.0)
    head.invalidate_cache()
    sd = {n: v.clone() for n, v in head.state_dict().items()}
    ocfg = oracle_cfg(cfg)
    taps = {}
    with torch.no_grad():
        ref, aux = O.focal_decoder_forward(sd, ocfg, inputs, taps)
        ocfg.mask_heatmap_mode = 'poscls'
        ref_pos, _ = O.focal_decoder_forward(sd, ocfg, inputs)
    for st in taps['stages']:
        v = torch.sort(st['heat'].reshape(1, -1), descending=True).values
        if not ((v[:, k - 1] - v[:, k]) > 1e-6).all():
            pytest.skip('seeded case has a top-k near-tie')
    extra_blank = int((ref['multistage_masks'][2] != ref_pos['multistage_masks'][2]).sum())
    assert extra_blank > 1000, extra_blank
    head = head.cuda()
    out = head(_cuda(inputs), None, [{}])[0][0]
    for m, r in zip(out['multistage_masks'], ref['multistage_masks']):
        assert torch.equal(m.cpu(), r), 'masks must be bit-exact'
    host = {key: v.cpu() for key, v in out.items() if torch.is_tensor(v)}
    perm = align_queries(host, ref, head.query_labels, aux['query_labels'], nq, k)
    assert torch.equal(head.query_labels.cpu(), permute_queries(aux['query_labels'], perm, nq)), 'query labels bit-exact'
    for key in ('center', 'height', 'dim', 'rot', 'vel', 'heatmap'):
        assert torch.allclose(host[key], permute_queries(ref[key], perm, nq), atol=1e-4, rtol=1e-4), key
    assert torch.allclose(host['query_box'], ref['query_box'].gather(2, perm[:, None, :].expand(-1, 10, -1)), atol=1e-4, rtol=1e-4)
