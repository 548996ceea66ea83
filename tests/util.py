"""Shared helpers for the parity tests: golden-fixture loading and index-set matching."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    sd, inp, out = {}, {}, {}
    for k in z.files:
        v = z[k]
        if k.startswith('sd/'):
            sd[k[3:]] = torch.from_numpy(v)
        elif k.startswith('sdtile/'):         # a large weight stored as its (41, 37, ...) channel tile (gen_golden.py: gen_neck_lss)
            shape = [int(n) for n in z['sdshape/' + k[7:]]]
            o, i = torch.arange(shape[0]) % v.shape[0], torch.arange(shape[1]) % v.shape[1]
            sd[k[7:]] = torch.from_numpy(v)[o][:, i].contiguous()
        elif k.startswith('in/'):
            inp[k[3:]] = torch.from_numpy(v)
        elif k.startswith('out/'):
            out[k[4:]] = torch.from_numpy(v)
    cfg = json.loads(bytes(z['cfg']).decode()) if 'cfg' in z.files else None
    return cfg, sd, inp, out, z


def oracle_cfg(cfg):
    from oracle import ff3d_oracle as O
    keys = ('num_proposals hidden_channel num_classes num_decoder_layers num_heads nms_kernel_size multiscale '
            'multistage_heatmap reuse_first_heatmap extra_feat bevpos input_img iterbev_wo_img mask_heatmap_mode '
            'roi_feats roi_expand_ratio roi_based_reg dataset pc_range voxel_size out_size_factor '
            'post_center_range score_threshold').split()
    kw = {k: cfg[k] for k in keys}
    kw['common_heads'] = {k: tuple(v) for k, v in cfg['common_heads'].items()}
    oc = O.head_config(**kw)
    oc.classaware_reg = bool(cfg.get('classaware_reg', False))
    oc.num_levels = cfg.get('num_levels', 3)
    oc.heatmap_box = bool(cfg.get('heatmap_box', False))
    oc.thin_heatmap_box = bool(cfg.get('thin_heatmap_box', False))
    return oc


def head_inputs(cfg, inp):
    n = sum(1 for k in inp if k.startswith('stage_'))
    maps = [inp[f'stage_{i}'] for i in range(n)]
    second = maps if cfg['multistage_heatmap'] else maps[0]
    return [inp['pts_feat_conv'], second]


def dense_pairs(out, ref):
    """(ours, reference) dense heatmaps: a list per heatmap head, or - single-stage branch without the second heatmap, FD:550-556 -
    ONE tensor, exactly as the reference returns it."""
    if 'dense_heatmap' in ref:
        assert torch.is_tensor(out['dense_heatmap']), 'the reference returns a tensor here, not a list'
        return [(out['dense_heatmap'], ref['dense_heatmap'])]
    assert isinstance(out['dense_heatmap'], (list, tuple))
    return [(h, ref[f'dense_heatmap/{i}']) for i, h in enumerate(out['dense_heatmap'])]


def stage_perm(ref_idx, our_idx):
    """Permutation p with ref_idx[:, p] == our_idx per row, asserting equal index SETS
    (the reference's top-k order is implementation-defined, FD:688)."""
    perms = []
    for r, o in zip(ref_idx, our_idx):
        assert torch.equal(torch.sort(r).values, torch.sort(o).values), 'top-k index sets differ'
        pos = {int(v): i for i, v in enumerate(r.tolist())}
        perms.append(torch.tensor([pos[int(v)] for v in o.tolist()]))
    return torch.stack(perms)


def head_kwargs(cfg):
    """Reference-style ``pts_bbox_head`` config dict (FocalFormer3D_L.py:238-314 layout) for a fixture cfg."""
    C = cfg['hidden_channel']
    dec = dict(type='DeformableDetrTransformerDecoder', num_layers=3, return_intermediate=False,
               transformerlayers=dict(
                   type='DetrTransformerDecoderLayer',
                   attn_cfgs=[dict(type='MultiheadAttention', embed_dims=C, num_heads=8, dropout=0.1),
                              dict(type='MultiScaleDeformableAttention', embed_dims=C,
                                   num_levels=cfg.get('num_levels', 3), num_points=4,
                                   num_heads=8)],
                   feedforward_channels=cfg.get('ffn_channels', 1024), ffn_dropout=0.1,
                   ffn_cfgs=dict(type='FFN', embed_dims=C, num_fcs=2, act_cfg=dict(type='ReLU', inplace=True)),
                   operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))
    Hb = cfg['grid']
    return dict(
        type='FocalDecoder', reuse_first_heatmap=cfg['reuse_first_heatmap'], extra_feat=cfg['extra_feat'],
        roi_feats=cfg['roi_feats'], roi_dropout_rate=0.1 if cfg['roi_feats'] else 0., roi_based_reg=cfg['roi_based_reg'],
        roi_expand_ratio=cfg['roi_expand_ratio'], hidden_channel_roi=cfg.get('hidden_channel_roi', 512),
        multiscale=cfg['multiscale'], multistage_heatmap=cfg['multistage_heatmap'] or None,
        mask_heatmap_mode=cfg['mask_heatmap_mode'], classaware_reg=cfg.get('classaware_reg', False),
        heatmap_box=cfg.get('heatmap_box', False), thin_heatmap_box=cfg.get('thin_heatmap_box', False), input_img=cfg['input_img'], iterbev_wo_img=cfg['iterbev_wo_img'],
        bevpos=cfg['bevpos'], num_proposals=cfg['num_proposals'], hidden_channel=C, num_classes=cfg['num_classes'],
        num_decoder_layers=cfg['num_decoder_layers'], num_heads=8, initialize_by_heatmap=True,
        nms_kernel_size=cfg['nms_kernel_size'], common_heads={k: tuple(v) for k, v in cfg['common_heads'].items()},
        bbox_coder=dict(type='TransFusionBBoxCoder', pc_range=cfg['pc_range'], voxel_size=cfg['voxel_size'],
                        out_size_factor=cfg['out_size_factor'], post_center_range=cfg['post_center_range'],
                        score_threshold=cfg['score_threshold'], code_size=10 if 'vel' in cfg['common_heads'] else 8),
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2, alpha=0.25, reduction='mean', loss_weight=1.0),
        decoder_cfg=dec,
        test_cfg=dict(dataset=cfg['dataset'], grid_size=[Hb * 8, Hb * 8, 40], out_size_factor=cfg['out_size_factor'],
                      pc_range=cfg['pc_range'], voxel_size=cfg['voxel_size'], nms_type=None))


class Boxes:
    """Stand-in for img_metas['box_type_3d'] (mmdet3d LiDARInstance3DBoxes is a plain container here)."""

    def __init__(self, tensor, box_dim=7, **kw):
        self.tensor, self.box_dim = tensor, box_dim


def oracle_cfg_from_head_cfg(hc):
    """oracle head_config from a reference-style ``pts_bbox_head`` dict (focalformer3d_amd.synthetic)."""
    from oracle import ff3d_oracle as O
    coder = hc['bbox_coder']
    return O.head_config(
        num_proposals=hc['num_proposals'], hidden_channel=hc['hidden_channel'], num_classes=hc['num_classes'],
        num_decoder_layers=hc['num_decoder_layers'], num_heads=hc['num_heads'], nms_kernel_size=hc['nms_kernel_size'],
        multiscale=hc['multiscale'], multistage_heatmap=hc['multistage_heatmap'] or 0,
        reuse_first_heatmap=hc['reuse_first_heatmap'], extra_feat=hc['extra_feat'], bevpos=hc['bevpos'],
        input_img=hc['input_img'], iterbev_wo_img=hc['iterbev_wo_img'], mask_heatmap_mode=hc['mask_heatmap_mode'],
        roi_feats=hc['roi_feats'], roi_expand_ratio=hc['roi_expand_ratio'], roi_based_reg=hc['roi_based_reg'],
        common_heads=hc['common_heads'], dataset=hc['test_cfg']['dataset'], pc_range=tuple(coder['pc_range']),
        voxel_size=tuple(coder['voxel_size']), out_size_factor=coder['out_size_factor'],
        post_center_range=tuple(coder['post_center_range']), score_threshold=coder['score_threshold'])


def align_queries(out, ref, labels, ref_labels, nq, k, max_moved=6):
    """Permutation of the reference's queries onto ours, per frame and HIP stage segment.

    Both sides order a stage's top-k by (score desc, lowest index); two candidates whose scores agree to fp32 round-off
    may swap ranks between two implementations (both are still selected, the k-th / (k+1)-th margin is checked
    separately), and everything downstream then differs only by that permutation.  Queries are matched on (label, first
    decoder stage centre); at most ``max_moved`` queries per frame may sit at a different rank.
    out / ref: dicts with 'center' (B, 2, D*nq); labels (B, nq).  Returns perm (B, nq) with ref[..., perm] ~ out."""
    from scipy.optimize import linear_sum_assignment
    ca, cb = out['center'][:, :, :nq].cpu().double(), ref['center'][:, :, :nq].cpu().double()
    la, lb = labels.cpu(), ref_labels.cpu()
    B = ca.shape[0]
    perm = torch.arange(nq).repeat(B, 1)
    for b in range(B):
        moved = 0
        for s0 in range(0, nq, k):
            sl = slice(s0, s0 + k)
            if torch.equal(la[b, sl], lb[b, sl]) and (ca[b, :, sl] - cb[b, :, sl]).abs().max() < 1e-3:
                continue
            cost = torch.cdist(ca[b, :, sl].t(), cb[b, :, sl].t()) + 1e3 * (la[b, sl, None] != lb[b, None, sl]).double()
            r, c = linear_sum_assignment(cost.numpy())
            assert cost[r, c].max() < 1e-3, 'a query of ours has no counterpart in the reference'
            perm[b, sl] = torch.as_tensor(c) + s0
            moved += int((torch.as_tensor(c) != torch.arange(len(c))).sum())
        assert moved <= max_moved, f'{moved} queries at a different rank'
    return perm


def permute_queries(t, perm, nq):
    """Apply align_queries' permutation to a (B, n, D*nq) tensor (every decoder-stage block) or a (B, nq) tensor."""
    if t.dim() == 2:
        return t.gather(1, perm)
    D = t.shape[-1] // nq
    full = torch.cat([perm + d * nq for d in range(D)], 1)
    return t.gather(2, full[:, None, :].expand(-1, t.shape[1], -1))


def load_decoder_hf(tag):
    """tests/golden/decoder_hf_<tag>.npz (oracle/gen_golden.py:gen_decoder_hf): mmcv-layout decoder parameters, inputs in the
    reference's call convention (batch-first) and the outputs of HF ``transformers``' ``DeformableDetrDecoder`` holding exactly
    those parameters.  Returns (sd, t, oracle cfg, shapes)."""
    from oracle import ff3d_oracle as O
    z = np.load(os.path.join(GOLDEN, f'decoder_hf_{tag}.npz'))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd/')}
    t = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith('sd/')}
    shapes = [tuple(int(v) for v in s) for s in z['shapes']]
    n_layers = 1 + max(int(k.split('.')[1]) for k in sd)
    cfg = O.head_config(num_heads=int(z['heads']), num_levels=len(shapes), num_points=int(z['points']), num_layers=n_layers,
                        hidden_channel=t['query'].shape[-1])
    return sd, t, cfg, shapes
