"""CPU tests: the oracle (our restatement) against the golden vectors produced by the REFERENCE
code itself (oracle/gen_golden.py) and by the independent HF MSDA implementation."""
import numpy as np
import pytest
import torch

from oracle import ff3d_oracle as O
from tests.util import dense_pairs, head_inputs, load_decoder_hf, load_golden, oracle_cfg, stage_perm

HEADS = ['head_focal_L', 'head_focal_LC', 'head_deform_L', 'head_waymo',
         'head_opt_classaware', 'head_opt_posmask', 'head_opt_singlescale', 'head_opt_singleheat',
         'head_opt_heatbox', 'head_opt_boxcls']


def test_posembed_matches_reference():
    _, sd, _, _, z = load_golden('posembed')
    pos = torch.from_numpy(z['pos'])
    emb = O.gen_sineembed_for_position(pos)
    assert torch.allclose(emb, torch.from_numpy(z['emb']), atol=1e-6, rtol=0)
    y = O.mlp(emb, sd, '')
    assert torch.allclose(y, torch.from_numpy(z['mlp']), atol=1e-5, rtol=1e-5)


def test_bbox_coder_matches_reference():
    z = np.load('tests/golden/bbox_coder.npz')
    t = {k: torch.from_numpy(z[k]) for k in z.files}
    cfg = O.head_config()
    box = O.decode_box(t['rot'], t['dim'], t['center'], t['height'], t['vel'], cfg)
    assert torch.allclose(box, t['decode_box'], atol=1e-5, rtol=1e-6)
    dicts, _ = O.bbox_decode(t['heat'], t['rot'], t['dim'], t['center'], t['height'], t['vel'], cfg)
    for i, d in enumerate(dicts):
        assert d['bboxes'].shape == t[f'bboxes{i}'].shape
        assert torch.allclose(d['bboxes'], t[f'bboxes{i}'], atol=1e-5, rtol=1e-6)
        assert torch.equal(d['scores'], t[f'scores{i}'])
        nz = d['scores'] > 0          # all-zero score columns: label is implementation-defined
        assert torch.equal(d['labels'][nz], t[f'labels{i}'][nz])


def test_bbox_coder_truthy_score_threshold_matches_reference():
    """BC:126-127, 140-141 with a threshold that is not 0.0: strict '>', a frame with no surviving box (empty result), no velocity."""
    z = np.load('tests/golden/bbox_coder_thr.npz')
    t = {k: torch.from_numpy(z[k]) for k in z.files if z[k].ndim}
    cfg = O.head_config(dataset='Waymo', pc_range=(-75.2, -75.2), voxel_size=(0.1, 0.1), score_threshold=float(z['score_threshold']),
                        post_center_range=(-80, -80, -10.0, 80, 80, 10.0), num_classes=3,
                        common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2)))
    dicts, _ = O.bbox_decode(t['heat'], t['rot'], t['dim'], t['center'], t['height'], None, cfg)
    assert [len(d['scores']) for d in dicts] == [34, 29, 0]
    for i, d in enumerate(dicts):
        assert d['bboxes'].shape == t[f'bboxes{i}'].shape and d['bboxes'].shape[1] == 7
        assert torch.allclose(d['bboxes'], t[f'bboxes{i}'], atol=1e-5, rtol=1e-6)
        assert torch.equal(d['scores'], t[f'scores{i}']) and torch.equal(d['labels'], t[f'labels{i}'])
        assert (d['scores'] > 0.35).all()
    cfg.score_threshold = 0.0                                   # the shipped value: falsy -> no score filter at all
    dicts0, _ = O.bbox_decode(t['heat'], t['rot'], t['dim'], t['center'], t['height'], None, cfg)
    assert len(dicts0[2]['scores']) > 0


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_msda_core_matches_hf(tag):
    z = np.load(f'tests/golden/msda_core_{tag}.npz')
    shapes = [tuple(int(v) for v in s) for s in z['shapes']]
    value, loc, w = (torch.from_numpy(z[k]) for k in ('value', 'loc', 'w'))
    ref = torch.from_numpy(z['out'])
    assert torch.allclose(O.msda_core(value, shapes, loc, w), ref, atol=2e-6, rtol=1e-5)
    # the CUDA-kernel formulation (explicit loops, fp64) agrees with the grid_sample formulation
    assert torch.allclose(O.msda_core_loops(value, shapes, loc, w), ref, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_decoder_sequence_and_layer_match_hf(tag):
    """Rows a13-a15 pinned by execution: the restated mmdet ``DeformableDetrTransformerDecoder`` / mmcv ``BaseTransformerLayer``
    wiring (q = k = x + pos, v = x, residual on the pre-pos query, post-norm order, reference point x valid ratio per level,
    the bool self-attention mask of FD:851-856) against HF ``DeformableDetrDecoder`` / ``DeformableDetrDecoderLayer`` holding the
    same parameters (measured: <= 3e-6)."""
    sd, t, cfg, shapes = load_decoder_hf(tag)
    q, pos, val = (t[k].transpose(0, 1) for k in ('query', 'query_pos', 'value'))        # sequence-first, as FD:927-933 passes them
    mask = None
    if 'attn_mask' in t:                                                                  # FD:856: (B * heads, Nq, Nq) bool
        mask = t['attn_mask'][:, None].repeat(1, cfg.num_heads, 1, 1).flatten(0, 1)
    taps = []
    out, ref_back = O.deformable_decoder(q, val, pos, t['reference_points'], shapes, t['valid_ratios'], sd, '', cfg,
                                         attn_mask=mask, taps=taps)
    assert ref_back is t['reference_points']
    assert len(taps) == t['per_layer'].shape[1]
    for l, tap in enumerate(taps):
        assert torch.allclose(tap.transpose(0, 1), t['per_layer'][:, l], atol=1e-5, rtol=1e-5), (l, float((tap.transpose(0, 1) - t['per_layer'][:, l]).abs().max()))
    assert torch.allclose(out.transpose(0, 1), t['out'], atol=1e-5, rtol=1e-5)
    ref_in = t['reference_points'][:, :, None] * t['valid_ratios'][:, None]
    one = O.decoder_layer(q, val, pos, ref_in, shapes, sd, 'layers.0.', cfg, attn_mask=mask)
    assert torch.allclose(one.transpose(0, 1), t['layer0'], atol=1e-5, rtol=1e-5)
    # the samples do leave the maps (the zero-padding branch is exercised), and case b's ratios differ from 1
    if tag == 'b':
        assert float((t['valid_ratios'] - 1).abs().min()) > 0


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_i2p_matches_reference(tag):
    _, sd, _, _, z = load_golden(f'i2p_{tag}')
    aug = torch.from_numpy(z['img_aug']) if 'img_aug' in z.files else None
    out = O.i2p_forward(sd, torch.from_numpy(z['lidar']), torch.from_numpy(z['img']),
                        torch.from_numpy(z['lidar2img']), tuple(int(v) for v in z['input_shape']), int(z['Z']),
                        img_aug=aug)
    ref = torch.from_numpy(z['out'])
    assert torch.equal(out.abs().sum(1) > 0, ref.abs().sum(1) > 0)
    assert torch.allclose(out, ref, atol=2e-6, rtol=1e-5)


def _margins(heat_flat, k):
    v = torch.sort(heat_flat, dim=-1, descending=True).values
    return v[:, k - 1] - v[:, k]


@pytest.mark.parametrize('name', HEADS)
def test_head_forward_matches_reference(name):
    cfg, sd, inp, ref, _ = load_golden(name)
    ocfg = oracle_cfg(cfg)
    taps = {}
    with torch.no_grad():
        out, aux = O.focal_decoder_forward(sd, ocfg, head_inputs(cfg, inp), taps)
    k = cfg['num_proposals']
    n_st = max(ocfg.num_stages, 1)
    nq = k * n_st
    assert aux['num_proposals'] == nq
    perms = []
    for i in range(n_st):
        st = taps['stages'][i]
        B = st['idx'].shape[0]
        assert (_margins(st['heat'].reshape(B, -1), k) > 1e-6).all(), 'fixture has a top-k tie'
        perms.append(stage_perm(ref[f'topk/{i}'][:, :k], st['idx']) + i * k)
    perm = torch.cat(perms, 1)                                   # our slot j == reference slot perm[j]
    B = perm.shape[0]
    assert torch.equal(ref['query_labels'].gather(1, perm), aux['query_labels'])
    qs = ref['query_heatmap_score'].gather(2, perm[:, None, :].expand(-1, cfg['num_classes'], -1))
    assert torch.allclose(out['query_heatmap_score'], qs, atol=1e-6, rtol=0)
    D = cfg['num_decoder_layers']
    full = torch.cat([perm + d * nq for d in range(D)], 1)
    for key in list(cfg['common_heads'].keys()) + ['heatmap']:
        r = ref[key].gather(2, full[:, None, :].expand(-1, ref[key].shape[1], -1))
        assert torch.allclose(out[key], r, atol=2e-5, rtol=1e-5), key
    for h, r in dense_pairs(out, ref):
        assert torch.allclose(h, r, atol=1e-5, rtol=1e-5)
    for i, m in enumerate(out.get('multistage_masks', [])):
        assert torch.equal(m.to(torch.uint8), ref[f'multistage_masks/{i}'])
    if cfg.get('heatmap_box'):                                   # FD:988-991: the branch's extra outputs
        for i, t in enumerate(out['multistage_bev_preds']):
            assert torch.allclose(t, ref[f'multistage_bev_preds/{i}'], atol=1e-5, rtol=1e-5)
        assert torch.allclose(out['query_box'], ref['query_box'].gather(2, perm[:, None, :].expand(-1, 10, -1)), atol=2e-5, rtol=1e-5)
        assert torch.allclose(out['query_pos'], ref['query_pos'].gather(1, perm[:, :, None].expand(-1, -1, 2)), atol=2e-5, rtol=1e-5)
    # RoI grid + sampled matrix as recorded from the reference's own grid_sample calls
    if cfg['roi_feats']:
        L = 3
        assert len(taps['roi_grid']) == (D if cfg.get('heatmap_box') else D - 1)      # heatmap boxes: RoI features from stage 0 on (FD:890)
        for s, (grid, mat) in enumerate(zip(taps['roi_grid'], taps['roi_mat'])):
            rg = ref[f'roi_grid/{s * L}'].gather(1, perm[:, :, None, None].expand(-1, -1, *grid.shape[2:]))
            assert torch.allclose(grid, rg, atol=1e-5, rtol=0)
            samp = torch.cat([ref[f'roi_sampled/{s * L + l}'] for l in range(L)], 1)     # (B,L*C,Nq,g*g)
            samp = samp.gather(2, perm[:, None, :, None].expand(-1, samp.shape[1], -1, samp.shape[3]))
            assert torch.allclose(mat, samp.permute(0, 2, 1, 3).reshape(mat.shape), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('name', HEADS)
def test_get_bboxes_matches_reference(name):
    cfg, sd, inp, ref, _ = load_golden(name)
    ocfg = oracle_cfg(cfg)
    first = [t[:1] for t in head_inputs(cfg, inp)[1]] if cfg['multistage_heatmap'] else head_inputs(cfg, inp)[1][:1]
    with torch.no_grad():
        out, aux = O.focal_decoder_forward(sd, ocfg, [inp['pts_feat_conv'][:1], first])
        (boxes, scores, labels), = O.focal_decoder_get_bboxes(out, aux, ocfg)[0]
    rb, rs, rl = ref['bboxes0'], ref['scores0'], ref['labels0']
    assert boxes.shape == rb.shape
    # order-independent match: sort both by (x, y)
    def order(b):
        return np.lexsort((b[:, 1].numpy(), b[:, 0].numpy()))
    a, b = order(boxes), order(rb)
    assert torch.allclose(boxes[a], rb[b], atol=1e-4, rtol=1e-5)
    assert torch.allclose(scores[a], rs[b], atol=1e-6, rtol=1e-5)
    nz = rs[b] > 0
    assert torch.equal(labels[a][nz], rl[b][nz])


@pytest.mark.parametrize('name', ['neck_mb2_lidar', 'neck_bevfusion_cam', 'neck_bevfusion_lss'])
def test_neck_matches_reference(name):
    """FocalEncoder oracle vs the golden produced by the reference's FocalEncoder source (its torchvision blocks and the
    CUDA-only locatt extension served by restatements - see oracle/gen_golden.py:gen_neck)."""
    cfg, sd, inp, ref, z = load_golden(name)
    l2i = inp.get('lidar2img')
    shape = tuple(int(v) for v in inp['input_shape']) if 'input_shape' in inp else None
    with torch.no_grad():
        new_img, (pts_conv, stages) = O.focal_encoder_forward(sd, cfg, inp.get('img_feats'), inp['pts_feats'], l2i, shape)
    assert torch.allclose(pts_conv, ref['pts_feat_conv'], atol=1e-5, rtol=1e-5)
    assert len(stages) == 3
    for i, t in enumerate(stages):
        assert torch.allclose(t, ref[f'stage_{i}'], atol=2e-5, rtol=1e-4), i
    if 'new_img_feat' in ref:
        assert torch.allclose(new_img, ref['new_img_feat'], atol=2e-5, rtol=1e-4)


def test_lss_matches_reference():
    """LiftSplatShoot oracle vs the golden produced by the reference's lss.py on CPU."""
    cfg, sd, inp, ref, _ = load_golden('lss_small')
    with torch.no_grad():
        bev, depth = O.lss_forward(sd, cfg, inp['x'], inp['rots'], inp['trans'])
    assert torch.allclose(depth, ref['depth'], atol=1e-6, rtol=1e-5)
    assert torch.allclose(bev, ref['bev'], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize('dataset', ['nuscenes', 'waymo'])
def test_get_bboxes_nms_variants_match_reference(dataset):
    """FD:1313-1413 with nms_type None / 'circle' / 'rotate' as the reference executes it (task tables, masks, keep indices, the
    200-box cap, pre / post sizes of the rotated NMS; oracle/gen_golden.py:gen_get_bboxes_nms) vs the oracle's restatement."""
    import json
    z = np.load(f'tests/golden/get_bboxes_nms_{dataset}.npz')
    c = json.loads(bytes(z['cfg']).decode())
    cfg = O.head_config(dataset=c['dataset'], num_classes=c['num_classes'], pc_range=tuple(c['pc_range']),
                        voxel_size=tuple(c['voxel_size']), out_size_factor=c['out_size_factor'],
                        post_center_range=tuple(c['post_center_range']), score_threshold=c['score_threshold'],
                        common_heads={k: tuple(v) for k, v in c['common_heads'].items()})
    out = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('in/') and k != 'in/query_labels'}
    ql = torch.from_numpy(z['in/query_labels'])
    n, K = ql.shape[1], c['num_classes']
    (b0, s0, l0), = O.focal_decoder_get_bboxes(out, dict(query_labels=ql, num_proposals=n), cfg)[0]
    score = out['heatmap'].sigmoid() * out['query_heatmap_score'] * torch.nn.functional.one_hot(ql, K).permute(0, 2, 1)
    dicts, _ = O.bbox_decode(score, out['rot'].clone(), out['dim'].clone(), out['center'].clone(), out['height'].clone(),
                             out['vel'].clone() if 'vel' in out else None, cfg)
    (b1, s1, l1), = O.get_bboxes_circle_nms(dicts, cfg)
    (b2, s2, l2), = O.get_bboxes_rotate_nms(dicts, cfg, c['pre_maxsize'], c['post_maxsize'])
    for tag, (b, s, l) in dict(none=(b0, s0, l0), circle=(b1, s1, l1), rotate=(b2, s2, l2)).items():
        rb, rs, rl = (torch.from_numpy(z[f'out/{tag}/{k}']) for k in ('bboxes', 'scores', 'labels'))
        assert b.shape == rb.shape, (tag, b.shape, rb.shape)
        assert torch.allclose(b, rb, atol=1e-5, rtol=1e-6) and torch.equal(s, rs) and torch.equal(l.int(), rl.int()), tag
    assert len(s0) == 200 and len(s1) <= 200 and len(s2) < 200 and len(dicts[0]["scores"]) == 300


def _merge_golden():
    import numpy as np
    import os
    from tests.util import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'merge_augs.npz'))
    n = sum(1 for k in z.files if k.startswith('in/boxes_'))
    augs = [dict(boxes=torch.from_numpy(z[f'in/boxes_{i}']), scores=torch.from_numpy(z[f'in/scores_{i}']),
                 labels=torch.from_numpy(z[f'in/labels_{i}']), scale=float(z[f'in/aug_{i}'][0]),
                 fh=bool(z[f'in/aug_{i}'][1]), fv=bool(z[f'in/aug_{i}'][2])) for i in range(n)]
    return augs, torch.from_numpy(z['out/boxes']), torch.from_numpy(z['out/scores']), torch.from_numpy(z['out/labels'])


def test_merge_aug_bboxes_matches_reference():
    """TTA merge oracle vs the golden produced by the reference's own ``merge_aug_bboxes_3d`` (core/post_processing/
    merge_augs.py:13-184; mapping back, per-class rotated NMS at 0.1, IoU voting at 0.65, top 500) - gen_golden.gen_merge_augs."""
    augs, rb, rs, rl = _merge_golden()
    b = torch.cat([O.bbox3d_mapping_back(a['boxes'], a['scale'], a['fh'], a['fv']) for a in augs])
    ob, os_, ol = O.merge_aug_boxes(b, torch.cat([a['scores'] for a in augs]), torch.cat([a['labels'] for a in augs]))
    assert ob.shape == rb.shape and 0 < len(ob) < len(b) // 2
    assert torch.equal(ol, rl) and torch.equal(os_, rs)
    assert torch.allclose(ob, rb, atol=2e-5, rtol=1e-5)


def load_train_golden():
    """tests/golden/train_targets.npz (oracle/gen_golden.py:gen_train): predictions of the reference head, ground truth,
    and the targets / losses the REFERENCE's get_targets / loss / HungarianAssigner3D produced for them."""
    import json
    import numpy as np
    import os
    from tests.util import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'train_targets.npz'))
    cfg = json.loads(bytes(z['cfg']).decode())
    preds = {}
    for key in z.files:
        if key.startswith('pred/'):
            parts = key.split('/')
            if len(parts) == 2:
                preds[parts[1]] = torch.from_numpy(z[key])
            else:
                preds.setdefault(parts[1], {})[int(parts[2])] = torch.from_numpy(z[key])
    for key, v in list(preds.items()):
        if isinstance(v, dict):
            preds[key] = [v[i] for i in range(len(v))]
    B = sum(1 for key in z.files if key.startswith('in/gt_boxes_'))
    gts = [torch.from_numpy(z[f'in/gt_boxes_{b}']) for b in range(B)]
    labels = [torch.from_numpy(z[f'in/gt_labels_{b}']) for b in range(B)]
    out = {key[4:]: torch.from_numpy(np.asarray(z[key])) for key in z.files if key.startswith('out/')}
    losses = {key[5:]: float(z[key]) for key in z.files if key.startswith('loss/')}
    return cfg, z, preds, gts, labels, out, losses


def test_training_targets_and_losses_match_reference():
    """HungarianAssigner3D + get_targets + loss: the oracle's restatement (oracle/train_oracle.py) vs what the reference's
    own code produced (FD:994-1311, hungarian_assigner.py:97-162)."""
    from oracle import train_oracle as T
    cfg, z, preds, gts, labels, ref, ref_losses = load_train_golden()
    h = cfg['head']
    ocfg = O.head_config(num_proposals=h['num_proposals'], hidden_channel=h['hidden_channel'], num_classes=h['num_classes'],
                         num_decoder_layers=h['num_decoder_layers'], pc_range=tuple(h['pc_range']),
                         voxel_size=tuple(h['voxel_size']), out_size_factor=h['out_size_factor'],
                         post_center_range=tuple(h['post_center_range']), score_threshold=h['score_threshold'])
    nq = h['num_proposals'] * 3
    kw = dict(num_proposals=nq, num_decoder_layers=h['num_decoder_layers'], num_classes=h['num_classes'], code_size=10,
              gt_center_limit=h['gt_center_limit'])
    got = T.get_targets(gts, labels, preds, ocfg, cfg['train_cfg'], **kw)
    names = ('labels', 'label_weights', 'bbox_targets', 'bbox_weights', 'ious', 'num_pos', 'matched_ious', 'heatmap')
    for name, v in zip(names, got):
        r = ref[name]
        if torch.is_tensor(v):
            assert v.shape == r.shape, name
            assert torch.allclose(v.float(), r.float(), atol=1e-5, rtol=1e-5), name
        else:
            assert abs(float(v) - float(r)) < 1e-5, name
    assert int(got[5]) > 10 and float(got[7].max()) == 1.0
    losses = T.head_loss(gts, labels, preds, ocfg, cfg['train_cfg'], cfg['losses'], **kw)
    assert set(losses) == set(ref_losses)
    for name, v in losses.items():
        assert abs(float(v) - ref_losses[name]) <= 1e-5 * max(1.0, abs(ref_losses[name])), (name, float(v), ref_losses[name])


def test_heuristic_assigner_matches_reference():
    """oracle.train_oracle.heuristic_assign vs the reference's HeuristicAssigner3D.assign (hungarian_assigner.py:58-91) executed
    under the import shims (oracle/gen_golden.py:gen_heuristic_assigner): assignment indices and labels bit-exact, IoUs 1e-6."""
    import os
    from oracle import train_oracle as T
    from tests.util import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'heuristic_assigner.npz'))
    for i in range(3):
        t = lambda k: torch.from_numpy(z[f'c{i}/{k}'])
        ql = t('query_labels') if int(z[f'c{i}/aware']) else None
        inds, overlaps, labels = T.heuristic_assign(t('pred'), t('gt'), t('gt_labels'), ql, float(z[f'c{i}/dist_thre']))
        assert torch.equal(inds, t('gt_inds')) and torch.equal(labels, t('labels').float())
        assert torch.allclose(overlaps, t('max_overlaps'), atol=1e-6)
