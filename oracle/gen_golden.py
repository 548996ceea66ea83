"""Generate tests/golden/*.npz by running the REFERENCE code (imported from /root/reference,
unmodified, under ``oracle/ref_shims``) on seeded inputs.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs /root/reference and,
for the MSDA-core vector, HF ``transformers``); the fixtures it writes are data - inputs,
weights and the reference's outputs - and are what travels to the GPU box.

    python -m oracle.gen_golden            # (re)writes every fixture
"""
import contextlib
import copy
import io
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shims as S  # noqa: E402

OUT = os.environ.get('FF3D_GOLDEN_OUT', os.path.join(os.path.dirname(HERE), 'tests', 'golden'))   # override to regenerate elsewhere and diff


def decoder_cfg(C, ffn=64, L=3, P=4, heads=8):
    return dict(type='DeformableDetrTransformerDecoder', num_layers=3, return_intermediate=False,
                transformerlayers=dict(
                    type='DetrTransformerDecoderLayer',
                    attn_cfgs=[dict(type='MultiheadAttention', embed_dims=C, num_heads=heads, dropout=0.1),
                               dict(type='MultiScaleDeformableAttention', embed_dims=C, num_levels=L,
                                    num_points=P, num_heads=heads)],
                    feedforward_channels=ffn, ffn_dropout=0.1,
                    ffn_cfgs=dict(type='FFN', embed_dims=C, num_fcs=2, act_cfg=dict(type='ReLU', inplace=True)),
                    operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))


def randomize(module, g):
    """Non-degenerate random weights and BN statistics (default inits leave BN an identity and
    the MSDA attention logits zero)."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) * (0.7 / max(1, p[0].numel()) ** 0.5))
            elif n.endswith('weight'):      # BN / LN scales
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        for n, b in module.named_buffers():
            if n.endswith('running_mean'):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            if n.endswith('running_var'):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)


class Recorder:
    """Record the reference's own top-k / argsort indices and grid_sample calls while it runs."""

    def __init__(self):
        self.topk, self.grids = [], []

    def __enter__(self):
        self._topk, self._argsort, self._gs = torch.topk, torch.Tensor.argsort, F.grid_sample

        def topk(*a, **k):
            r = self._topk(*a, **k)
            self.topk.append(r.indices.clone())
            return r

        def argsort(t, *a, **k):
            r = self._argsort(t, *a, **k)
            if t.dim() == 2 and t.shape[-1] > 1000:
                self.topk.append(r.clone())
            return r

        def gs(inp, grid, *a, **k):
            r = self._gs(inp, grid, *a, **k)
            if sys._getframe(1).f_code.co_filename.endswith('focal_decoder.py'):   # the reference's own calls only
                self.grids.append((grid.clone(), r.clone()))
            return r
        torch.topk, torch.Tensor.argsort, F.grid_sample = topk, argsort, gs
        return self

    def __exit__(self, *e):
        torch.topk, torch.Tensor.argsort, F.grid_sample = self._topk, self._argsort, self._gs


def np_sd(sd):
    return {'sd/' + k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}


def gen_head(ref, name, seed, C, K, Hb, k, dataset, multistage, reuse, extra, roi, D, vel=True, B=2,
             input_img=False, iterbev_wo_img=True, classaware=False, mask_mode='poscls', multiscale=True, bevpos=True,
             heatmap_box=False, weight_gain=None):
    g = torch.Generator().manual_seed(seed)
    heads = dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2))
    if vel:
        heads['vel'] = (2, 2)
    nus = dataset == 'nuScenes'
    pcr = [-54.0, -54.0] if nus else [-75.2, -75.2]
    vox = 2 * abs(pcr[0]) / (Hb * 8)
    coder = dict(type='TransFusionBBoxCoder', pc_range=pcr, voxel_size=[vox, vox], out_size_factor=8,
                 post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0] if nus else [-80, -80, -10.0, 80, 80, 10.0],
                 score_threshold=0.0, code_size=10 if vel else 8)
    kw = dict(reuse_first_heatmap=reuse, extra_feat=extra, roi_feats=roi, roi_dropout_rate=0.1 if roi else 0.,
              roi_based_reg=bool(roi), roi_expand_ratio=1.2, hidden_channel_roi=48,
              multiscale=multiscale, multistage_heatmap=multistage, mask_heatmap_mode=mask_mode,
              classaware_reg=classaware, heatmap_box=heatmap_box, thin_heatmap_box=heatmap_box,
              input_img=input_img, iterbev_wo_img=iterbev_wo_img, bevpos=bevpos, num_proposals=k, hidden_channel=C,
              num_classes=K, num_decoder_layers=D, num_heads=8, initialize_by_heatmap=True, nms_kernel_size=3,
              common_heads=heads, bbox_coder=coder, loss_cls=dict(type='FocalLoss', use_sigmoid=True),
              decoder_cfg=decoder_cfg(C, L=3 if multiscale else 1),
              test_cfg=dict(dataset=dataset, grid_size=[Hb * 8, Hb * 8, 40], out_size_factor=8, pc_range=pcr,
                            voxel_size=[vox, vox], nms_type=None))
    head = ref.FocalDecoder(**kw).eval()
    randomize(head, g)
    if weight_gain:                                  # e.g. the task heads' last conv: boxes that cover more than their own cell
        with torch.no_grad():
            for n, v in head.named_parameters():
                for pat, gain in weight_gain.items():
                    if pat in n:
                        v.mul_(gain)
    sd = {n: v.clone() for n, v in head.state_dict().items()}
    n_maps = (multistage or 0) + (1 if extra else 0)
    f0 = torch.randn(B, C, Hb, Hb, generator=g)
    maps = [torch.randn(B, C, Hb, Hb, generator=g) for _ in range(max(n_maps, 1))]
    second = list(maps) if multistage else maps[0]
    data = dict(np_sd(sd))
    data['in/pts_feat_conv'] = f0.numpy()
    for i, m in enumerate(maps):
        data[f'in/stage_{i}'] = m.numpy()
    with torch.no_grad(), S.cpu_device_patch(), Recorder() as rec:
        out = head([f0.clone(), [m.clone() for m in second] if multistage else second.clone()], None, [{}] * B)[0][0]
    for key, v in out.items():
        if torch.is_tensor(v):
            data['out/' + key] = v.numpy()
        elif key == 'multistage_bev_preds':          # FD:988-989: per stage, per task a dict of views -> one (B, 6 * 10, H, W) array per stage
            for i, tasks in enumerate(v):
                data[f'out/{key}/{i}'] = torch.cat([torch.cat([t['reg'], t['height'], t['dim'], t['rot'], t['vel']], 1) for t in tasks], 1).numpy()
        else:
            for i, t in enumerate(v):
                data[f'out/{key}/{i}'] = t.numpy().astype(np.uint8) if key == 'multistage_masks' else t.numpy()
    data['out/query_labels'] = head.query_labels.numpy()
    for i, t in enumerate(rec.topk):
        data[f'out/topk/{i}'] = t.numpy()
    for i, (grid, samp) in enumerate(rec.grids):
        data[f'out/roi_grid/{i}'] = grid.numpy()
        data[f'out/roi_sampled/{i}'] = samp.numpy()
    # get_bboxes on sample 0 alone (the reference asserts batch == 1, FD:1406-1407)
    with torch.no_grad(), S.cpu_device_patch():
        sec1 = [m[:1].clone() for m in second] if multistage else second[:1].clone()
        o1 = head([f0[:1].clone(), sec1], None, [{}])
        boxes, scores, labels = head.get_bboxes(o1, [{'box_type_3d': S.LiDARInstance3DBoxes}])[0]
    data['out/bboxes0'] = boxes.tensor.numpy()
    data['out/scores0'] = scores.numpy()
    data['out/labels0'] = labels.numpy()
    cfg = dict(num_proposals=k, hidden_channel=C, num_classes=K, num_decoder_layers=D, num_heads=8,
               nms_kernel_size=3, multiscale=multiscale, multistage_heatmap=multistage or 0, reuse_first_heatmap=reuse,
               extra_feat=extra, bevpos=bevpos, input_img=input_img, iterbev_wo_img=iterbev_wo_img,
               mask_heatmap_mode=mask_mode, roi_feats=roi, roi_expand_ratio=1.2, roi_based_reg=bool(roi),
               common_heads={a: list(b) for a, b in heads.items()}, dataset=dataset,
               pc_range=pcr, voxel_size=[vox, vox], out_size_factor=8,
               post_center_range=coder['post_center_range'], score_threshold=0.0,
               hidden_channel_roi=48, ffn_channels=64, grid=Hb)
    if classaware:                                   # (keys the four config-shaped fixtures do not carry: tests default them)
        cfg['classaware_reg'] = True
    if not multiscale:
        cfg['num_levels'] = 1
    if heatmap_box:
        cfg['heatmap_box'] = cfg['thin_heatmap_box'] = True
    data['cfg'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **data)
    print(name, 'written;', sum(v.nbytes for v in data.values()) // 1024, 'KiB raw;',
          'boxes', tuple(boxes.tensor.shape))


def gen_posembed(ref):
    g = torch.Generator().manual_seed(7)
    pos = torch.rand(2, 16, 2, generator=g) * 1.2 - 0.1
    emb = ref.utils.gen_sineembed_for_position(pos)
    m = ref.utils.MLP(256, 24, 24, 2)
    randomize(m, g)
    with torch.no_grad():
        y = m(emb)
    data = dict(np_sd(m.state_dict()))
    data.update(pos=pos.numpy(), emb=emb.numpy(), mlp=y.numpy())
    np.savez_compressed(os.path.join(OUT, 'posembed.npz'), **data)
    print('posembed written')


def gen_coder(ref):
    g = torch.Generator().manual_seed(11)
    coder = ref.TransFusionBBoxCoder(pc_range=[-54.0, -54.0], out_size_factor=8, voxel_size=[0.075, 0.075],
                                     post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                                     score_threshold=0.0, code_size=10)
    B, K, N = 2, 10, 40
    heat = torch.rand(B, K, N, generator=g)
    heat[:, :, :5] = 0.0                                    # all-zero columns (label = don't care)
    rot = torch.randn(B, 2, N, generator=g)
    dim = torch.randn(B, 3, N, generator=g) * 0.5
    center = torch.rand(B, 2, N, generator=g) * 220 - 20   # some outside post_center_range
    height = torch.randn(B, 1, N, generator=g) * 6         # some |z| > 10
    vel = torch.randn(B, 2, N, generator=g)
    with torch.no_grad():
        res = coder.decode(heat.clone(), rot.clone(), dim.clone(), center.clone(), height.clone(), vel.clone(),
                           filter=True)
        box9 = coder.decode_box(rot.clone(), dim.clone(), center.clone(), height.clone(), vel.clone())
    data = dict(heat=heat.numpy(), rot=rot.numpy(), dim=dim.numpy(), center=center.numpy(), height=height.numpy(),
                vel=vel.numpy(), decode_box=box9.numpy())
    for i, r in enumerate(res):
        data[f'bboxes{i}'] = r['bboxes'].numpy()
        data[f'scores{i}'] = r['scores'].numpy()
        data[f'labels{i}'] = r['labels'].numpy()
    np.savez_compressed(os.path.join(OUT, 'bbox_coder.npz'), **data)
    print('bbox_coder written', [tuple(r['bboxes'].shape) for r in res])
    gen_coder_threshold(ref)


def gen_coder_threshold(ref):
    """decode(filter=True) with a TRUTHY score threshold (BC:126-127, 140-141 - the configs ship 0.0, which disables it) on three
    frames: mixed, everything kept by the threshold, and one where no score passes (an EMPTY result, shapes (0, 7) / (0,));
    Waymo-style code (no velocity, code_size 8)."""
    g = torch.Generator().manual_seed(12)
    coder = ref.TransFusionBBoxCoder(pc_range=[-75.2, -75.2], out_size_factor=8, voxel_size=[0.1, 0.1],
                                     post_center_range=[-80, -80, -10.0, 80, 80, 10.0], score_threshold=0.35, code_size=8)
    B, K, N = 3, 3, 48
    heat = torch.rand(B, K, N, generator=g)
    heat[1] = 0.4 + 0.6 * heat[1]                           # frame 1: every score above the threshold
    heat[2] = 0.35 * heat[2]                                # frame 2: none (0.35 itself would not pass either: strict >)
    heat[0, :, 7] = 0.35                                    # frame 0: a score exactly AT the threshold is dropped
    rot = torch.randn(B, 2, N, generator=g)
    dim = torch.randn(B, 3, N, generator=g) * 0.5
    center = torch.rand(B, 2, N, generator=g) * 230 - 20   # some outside post_center_range
    height = torch.randn(B, 1, N, generator=g) * 5
    with torch.no_grad():
        res = coder.decode(heat.clone(), rot.clone(), dim.clone(), center.clone(), height.clone(), None, filter=True)
        coder.post_center_range = [-80, -80, -10.0, 80, 80, 10.0]      # (decode() replaced the list by a tensor, BC:130-131)
        allq = coder.decode(heat.clone(), rot.clone(), dim.clone(), center.clone(), height.clone(), None, filter=False)
    data = dict(heat=heat.numpy(), rot=rot.numpy(), dim=dim.numpy(), center=center.numpy(), height=height.numpy(),
                score_threshold=np.float32(0.35))
    for i, (r, a) in enumerate(zip(res, allq)):
        data[f'bboxes{i}'], data[f'scores{i}'], data[f'labels{i}'] = r['bboxes'].numpy(), r['scores'].numpy(), r['labels'].numpy()
        data[f'all_bboxes{i}'], data[f'all_scores{i}'], data[f'all_labels{i}'] = (a['bboxes'].numpy(), a['scores'].numpy(),
                                                                               a['labels'].numpy())
    np.savez_compressed(os.path.join(OUT, 'bbox_coder_thr.npz'), **data)
    print('bbox_coder_thr written', [tuple(r['bboxes'].shape) for r in res])


def gen_msda_hf():
    """MSDA core pinned on the independent HF implementation of the same Deformable-DETR op."""
    from transformers.models.deformable_detr.modeling_deformable_detr import MultiScaleDeformableAttention
    g = torch.Generator().manual_seed(3)
    shapes = [(12, 12), (6, 6), (3, 3)]
    for tag, (B, Nq, M, D) in dict(a=(2, 17, 8, 16), b=(1, 9, 8, 32), c=(2, 5, 4, 8)).items():
        Nv = sum(h * w for h, w in shapes)
        value = torch.randn(B, Nv, M, D, generator=g)
        loc = torch.rand(B, Nq, M, 3, 4, 2, generator=g) * 1.5 - 0.25    # incl. out-of-range
        loc[0, 0, 0, 0, 0] = torch.tensor([0.0, 0.0])
        loc[0, 0, 0, 0, 1] = torch.tensor([1.0, 1.0])
        loc[0, 0, 0, 1, 0] = torch.tensor([-1.0 / 12, 0.5])               # exactly on the -1 pixel edge
        w = torch.rand(B, Nq, M, 12, generator=g).softmax(-1).view(B, Nq, M, 3, 4)
        out = MultiScaleDeformableAttention()(value, torch.tensor(shapes), shapes, None, loc, w, 64)
        np.savez_compressed(os.path.join(OUT, f'msda_core_{tag}.npz'), value=value.numpy(), loc=loc.numpy(),
                            w=w.numpy(), out=out.numpy(), shapes=np.array(shapes))
    print('msda_core (HF) written')


def hf_decoder_from_mmcv(sd, C, heads, L, P, ffn, n_layers):
    """HF ``transformers`` ``DeformableDetrDecoder`` (an independent implementation of the Deformable-DETR decoder the reference
    builds from mmcv / mmdet config strings, FocalFormer3D_L.py:285-313 -> FD:16,304) holding OUR mmcv-layout parameters
    (SURVEY Appendix B key names under ``layers.<l>.``).  The only arithmetic here is the key mapping."""
    from transformers import DeformableDetrConfig, ResNetConfig
    from transformers.models.deformable_detr.modeling_deformable_detr import DeformableDetrDecoder
    hf_cfg = DeformableDetrConfig(d_model=C, decoder_layers=n_layers, decoder_attention_heads=heads, decoder_ffn_dim=ffn,
                                  decoder_n_points=P, num_feature_levels=L, activation_function='relu', dropout=0.0,
                                  attention_dropout=0.0, activation_dropout=0.0, disable_custom_kernels=True,
                                  backbone_config=ResNetConfig(out_features=['stage4']),     # (never built: decoder only; avoids a hub lookup)
                                  encoder_layers=1, encoder_ffn_dim=ffn, encoder_attention_heads=heads, encoder_n_points=P)
    hf_cfg._attn_implementation = 'eager'
    dec = DeformableDetrDecoder(hf_cfg).eval()
    m = {}
    for l in range(n_layers):
        p, h = f'layers.{l}.', f'layers.{l}.'
        w, b = sd[p + 'attentions.0.attn.in_proj_weight'], sd[p + 'attentions.0.attn.in_proj_bias']
        for i, n in enumerate('qkv'):                                   # nn.MultiheadAttention packs q, k, v rows in this order
            m[h + f'self_attn.{n}_proj.weight'], m[h + f'self_attn.{n}_proj.bias'] = w[i * C:(i + 1) * C], b[i * C:(i + 1) * C]
        m[h + 'self_attn.o_proj.weight'] = sd[p + 'attentions.0.attn.out_proj.weight']
        m[h + 'self_attn.o_proj.bias'] = sd[p + 'attentions.0.attn.out_proj.bias']
        for n in ('sampling_offsets', 'attention_weights', 'value_proj', 'output_proj'):
            for t in ('weight', 'bias'):
                m[h + f'encoder_attn.{n}.{t}'] = sd[p + f'attentions.1.{n}.{t}']
        for t in ('weight', 'bias'):
            m[h + f'mlp.fc1.{t}'] = sd[p + f'ffns.0.layers.0.0.{t}']
            m[h + f'mlp.fc2.{t}'] = sd[p + f'ffns.0.layers.1.{t}']
            for ours, theirs in (('norms.0', 'self_attn_layer_norm'), ('norms.1', 'encoder_attn_layer_norm'),
                                 ('norms.2', 'final_layer_norm')):
                m[h + f'{theirs}.{t}'] = sd[p + f'{ours}.{t}']
    missing, unexpected = dec.load_state_dict(m, strict=True)
    assert not missing and not unexpected
    return dec


def gen_decoder_hf():
    """Rows a13-a15 (decoder sequence, decoder layer, MSDA module: mmdet ``DeformableDetrTransformerDecoder.forward``, mmcv
    ``BaseTransformerLayer.forward`` / ``MultiheadAttention`` / ``FFN`` / ``MultiScaleDeformableAttention.forward`` as driven at
    FD:927-933) pinned by EXECUTION of an independent implementation: HF ``DeformableDetrDecoder`` /
    ``DeformableDetrDecoderLayer`` loaded with mmcv-layout parameters.  Inputs in the reference's call convention
    (batch-first here; FD permutes to sequence-first), per-layer hidden states recorded."""
    g = torch.Generator().manual_seed(61)
    cases = dict(
        # FocalFormer3D_L-shaped: 3 layers, 3 levels, 4 points, 8 heads, valid_ratios = ones (FD:863)
        a=dict(C=64, heads=8, L=3, P=4, ffn=128, n_layers=3, B=2, Nq=50, shapes=[(20, 20), (10, 10), (5, 5)], ratios='ones',
               mask=False),
        # single-scale value (multiscale=False, FD:835-838), 1 layer, and valid ratios != 1 (the generic mmdet path)
        b=dict(C=32, heads=8, L=1, P=4, ffn=64, n_layers=1, B=3, Nq=21, shapes=[(12, 12)], ratios='random', mask=False),
        # the training-time self-attention mask of FD:851-856 (bool, True = may NOT attend) as HF's additive mask
        c=dict(C=32, heads=8, L=3, P=4, ffn=96, n_layers=2, B=2, Nq=24, shapes=[(16, 16), (8, 8), (4, 4)], ratios='ones',
               mask=True))
    for tag, c in cases.items():
        C, heads, L, P, B, Nq = c['C'], c['heads'], c['L'], c['P'], c['B'], c['Nq']
        sd = {}
        for l in range(c['n_layers']):          # mmcv-layout keys and shapes, non-degenerate random values
            p = f'layers.{l}.'
            lin_shapes = {'attentions.0.attn.in_proj_': (3 * C, C), 'attentions.0.attn.out_proj.': (C, C),
                          'attentions.1.sampling_offsets.': (heads * L * P * 2, C), 'attentions.1.attention_weights.': (heads * L * P, C),
                          'attentions.1.value_proj.': (C, C), 'attentions.1.output_proj.': (C, C),
                          'ffns.0.layers.0.0.': (c['ffn'], C), 'ffns.0.layers.1.': (C, c['ffn'])}
            for n, (o, i) in lin_shapes.items():
                sd[p + n + 'weight'] = torch.randn(o, i, generator=g) * (0.7 / i ** 0.5)
                sd[p + n + 'bias'] = torch.randn(o, generator=g) * 0.1
            for n in range(3):
                sd[p + f'norms.{n}.weight'] = 1 + 0.1 * torch.randn(C, generator=g)
                sd[p + f'norms.{n}.bias'] = 0.1 * torch.randn(C, generator=g)
        for l in range(c['n_layers']):          # offsets of several pixels: samples leave the maps on every side
            sd[f'layers.{l}.attentions.1.sampling_offsets.bias'] = torch.randn(heads * L * P * 2, generator=g) * 3.0
        hf = hf_decoder_from_mmcv(sd, C, heads, L, P, c['ffn'], c['n_layers'])
        shapes = c['shapes']
        Nv = sum(h * w for h, w in shapes)
        q, pos = torch.randn(B, Nq, C, generator=g), torch.randn(B, Nq, C, generator=g)
        val = torch.randn(B, Nv, C, generator=g)
        ref_pts = torch.rand(B, Nq, 2, generator=g) * 1.1 - 0.05
        ref_pts[0, 0] = torch.tensor([0.0, 0.0])
        ref_pts[0, 1] = torch.tensor([1.0, 1.0])
        ratios = torch.ones(B, L, 2) if c['ratios'] == 'ones' else 0.6 + 0.4 * torch.rand(B, L, 2, generator=g)
        kw = {}
        mask = None
        if c['mask']:
            n0 = Nq // 2                       # FD:851-856: everyone sees the first n0 queries; the rest see a random valid subset
            valid = torch.rand(B, Nq - n0, generator=g) > 0.4
            mask = torch.ones(B, Nq, Nq, dtype=torch.bool)
            mask[:, :, :n0] = False
            mask[:, n0:, n0:] = ~(valid[:, None] & valid[:, :, None])
            add = torch.zeros(B, 1, Nq, Nq).masked_fill(mask[:, None], float('-inf'))
            kw['attention_mask'] = add
        ss = torch.tensor(shapes)
        lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
        with torch.no_grad():
            out = hf(inputs_embeds=q, encoder_hidden_states=val, object_queries_position_embeddings=pos, reference_points=ref_pts,
                     spatial_shapes=ss, spatial_shapes_list=shapes, level_start_index=lsi, valid_ratios=ratios, **kw)
            # one layer on its own, driven with the per-level reference points directly (the a14 / a15 boundary)
            ref_in = ref_pts[:, :, None] * ratios[:, None]
            one = hf.layers[0](q, pos, ref_in, ss, shapes, lsi, val, None, **kw)
        data = {'sd/' + k: v.numpy() for k, v in sd.items()}
        data.update(query=q.numpy(), query_pos=pos.numpy(), value=val.numpy(), reference_points=ref_pts.numpy(),
                    valid_ratios=ratios.numpy(), shapes=np.array(shapes), heads=np.int64(heads), points=np.int64(P),
                    ffn=np.int64(c['ffn']), out=out.last_hidden_state.numpy(), per_layer=out.intermediate_hidden_states.numpy(),
                    layer0=one.numpy())
        if mask is not None:
            data['attn_mask'] = mask.numpy()
        assert torch.equal(out.intermediate_hidden_states[:, 0], one)
        np.savez_compressed(os.path.join(OUT, f'decoder_hf_{tag}.npz'), **data)
    print('decoder_hf (HF DeformableDetrDecoder) written')


def gen_i2p(ref):
    g = torch.Generator().manual_seed(5)
    for tag, (Cp, Ci, aug) in dict(a=(16, 16, False), b=(16, 24, True)).items():
        B, ncam, H, W, Z, Hi, Wi = 2, 3, 12, 12, 4, 8, 16
        m = ref.I2P(Cp, Ci, 0.1, max_points_height=Z).eval()
        randomize(m, g)
        lidar = torch.randn(B, Cp, H, W, generator=g)
        img = torch.randn(B, ncam, Ci, Hi, Wi, generator=g)
        input_shape = (Hi * 4, Wi * 4)
        # synthetic pinhole cameras looking outward at 120 deg spacing
        l2i = []
        for b in range(B):
            mats = []
            for c in range(ncam):
                yaw = 2 * np.pi * c / ncam + 0.3 * b
                fwd = np.array([np.cos(yaw), np.sin(yaw), 0.0])
                right = np.array([np.sin(yaw), -np.cos(yaw), 0.0])
                down = np.array([0.0, 0.0, -1.0])
                R = np.stack([right, down, fwd])                         # lidar -> camera axes
                t = -R @ np.array([0.5 * np.cos(yaw), 0.5 * np.sin(yaw), 1.0])
                f = 0.6 * input_shape[1]
                Kmat = np.array([[f, 0, input_shape[1] / 2], [0, f, input_shape[0] / 2], [0, 0, 1.0]])
                M = np.eye(4)
                M[:3, :3] = Kmat @ R
                M[:3, 3] = Kmat @ t
                mats.append(M)
            l2i.append(np.stack(mats))
        l2i = np.stack(l2i).astype(np.float32)
        metas = []
        augm = None
        if aug:
            augm = torch.eye(4).repeat(B, ncam, 1, 1)
            augm[..., 0, 0] = 0.9
            augm[..., 1, 1] = 0.9
            augm[..., 0, 3] = 2.0
            augm[..., 1, 3] = -1.5
        for b in range(B):
            meta = dict(lidar2img=l2i[b], input_shape=input_shape)
            if aug:
                meta['img_aug_matrix'] = augm[b]
            metas.append(meta)
        with torch.no_grad(), S.cpu_device_patch():
            out = m(lidar.clone(), img.clone(), metas)
        data = dict(np_sd(m.state_dict()))
        data.update(lidar=lidar.numpy(), img=img.numpy(), lidar2img=l2i, input_shape=np.array(input_shape),
                    Z=np.array(Z), out=out.numpy())
        if aug:
            data['img_aug'] = augm.numpy()
        np.savez_compressed(os.path.join(OUT, f'i2p_{tag}.npz'), **data)
        print('i2p', tag, 'written; nonzero pillars', int((out.abs().sum(1) > 0).sum()), 'of', B * H * W)


def synthetic_rig(B, ncam, input_shape, yaw0=0.0):
    l2i = []
    for b in range(B):
        mats = []
        for c in range(ncam):
            yaw = 2 * np.pi * c / ncam + 0.3 * b + yaw0
            fwd = np.array([np.cos(yaw), np.sin(yaw), 0.0])
            right = np.array([np.sin(yaw), -np.cos(yaw), 0.0])
            down = np.array([0.0, 0.0, -1.0])
            R = np.stack([right, down, fwd])
            t = -R @ np.array([0.5 * np.cos(yaw), 0.5 * np.sin(yaw), 1.0])
            f = 0.6 * input_shape[1]
            Kmat = np.array([[f, 0, input_shape[1] / 2], [0, f, input_shape[0] / 2], [0, 0, 1.0]])
            M = np.eye(4)
            M[:3, :3] = Kmat @ R
            M[:3, 3] = Kmat @ t
            mats.append(M)
        l2i.append(np.stack(mats))
    return np.stack(l2i).astype(np.float32)


def gen_neck(ref, name, seed, iterbev, with_img):
    """FocalEncoder (necks/focal_encoder.py) run from the reference source; its torchvision blocks and the CUDA-only
    locatt extension are served by the shims (oracle restatements), so this pins the reference's own glue code."""
    g = torch.Generator().manual_seed(seed)
    B, C, H, Cin_p, Cin_i, ncam, Hi, Wi, Z = 2, 16, 14, 24, 20, 3, 8, 16, 4
    m = ref.FocalEncoder(num_layers=2, in_channels_img=Cin_i, in_channels_pts=Cin_p, hidden_channel=C, iterbev=iterbev,
                         max_points_height=Z, multistage_heatmap=2, input_img=with_img, input_pts=True,
                         iterbev_wo_img=not with_img, extra_feat=True, iter_bev_cam=with_img, cam_lss=False).eval()
    randomize(m, g)
    pts = torch.randn(B, Cin_p, H, H, generator=g)
    data = dict(np_sd(m.state_dict()))
    data['in/pts_feats'] = pts.numpy()
    metas = [{} for _ in range(B)]
    img = None
    if with_img:
        img = torch.randn(B * ncam, Cin_i, Hi, Wi, generator=g)
        shape = (Hi * 4, Wi * 4)
        l2i = synthetic_rig(B, ncam, shape)
        metas = [dict(lidar2img=l2i[b], input_shape=shape) for b in range(B)]
        data.update({'in/img_feats': img.numpy(), 'in/lidar2img': l2i, 'in/input_shape': np.array(shape)})
    with torch.no_grad(), S.cpu_device_patch():
        new_img, (pts_conv, stages) = m(None if img is None else img.clone(), pts.clone(), metas)
    data['out/pts_feat_conv'] = pts_conv.numpy()
    for i, t in enumerate(stages):
        data[f'out/stage_{i}'] = t.numpy()
    if new_img is not None:
        data['out/new_img_feat'] = new_img.numpy()
    cfg = dict(num_layers=2, in_channels_img=Cin_i, in_channels_pts=Cin_p, hidden_channel=C, iterbev=iterbev,
               max_points_height=Z, multistage_heatmap=2, input_img=with_img, input_pts=True,
               iterbev_wo_img=not with_img, extra_feat=True, iter_bev_cam=with_img, cam_lss=False)
    data['cfg'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **data)
    print(name, 'written;', len(stages), 'stage maps')


def gen_neck_lss(ref):
    """FocalEncoder with the Lift-Splat-Shoot camera branch inside (cam_lss=True, iterbev='bevfusion', iter_bev_cam=True: the neck of
    FocalFormer3D_LC.py:190-200) run from the reference source: the lidar2img inversion per camera, LiftSplatShoot at its fixed
    widths (inputC 256, camC 64, grid 0.6 m) and the fusion blocks without their own projection (need_projbev=False).  A small
    range (36 x 36 cells of 0.6 m) keeps the fixed 512-wide BEV encoder cheap."""
    g = torch.Generator().manual_seed(33)
    pc = [-10.8, -10.8, -5.0, 10.8, 10.8, 3.0]
    B, C, H, Cin_p, ncam, Z, shape = 1, 16, 36, 24, 3, 4, (64, 112)
    cfg = dict(num_layers=2, in_channels_img=256, in_channels_pts=Cin_p, hidden_channel=C, iterbev='bevfusion',
               max_points_height=Z, multistage_heatmap=2, input_img=True, input_pts=True, iterbev_wo_img=False, extra_feat=True,
               iter_bev_cam=True, cam_lss=True, pc_range=pc, img_scale=shape)
    with S.cpu_device_patch():
        m = ref.FocalEncoder(**cfg).eval()
    randomize(m, g)
    with torch.no_grad():
        for prm in m.parameters():                                # the fixed-width LSS encoders, so that the fixture compresses:
            if prm.numel() > 200000:                              # 832 / 512-wide 3x3 convs (48 MB): a 41 x 37 channel tile
                o, i = torch.arange(prm.shape[0]) % 41, torch.arange(prm.shape[1]) % 37      # repeated (odd periods: a channel
                base = torch.randn(41, 37, 3, 3, generator=g) * 0.012                        # mix-up by a power of two shows)
                prm.copy_(base[o][:, i])
            elif prm.numel() > 20000:                             # 15-level weights
                prm.copy_(torch.randint(-7, 8, prm.shape, generator=g).float() * 0.005)
        m.cam_lss.frustum.copy_(m.cam_lss.create_frustum())     # (a Parameter too: neither randomize() nor the loop above may keep it)
    pts = torch.randn(B, Cin_p, H, H, generator=g)
    img = torch.randn(B * ncam, 256, shape[0] // 4, shape[1] // 4, generator=g)
    l2i = synthetic_rig(B, ncam, shape)
    metas = [dict(lidar2img=l2i[b], input_shape=shape) for b in range(B)]
    with torch.no_grad(), S.cpu_device_patch():
        new_img, (pts_conv, stages) = m(img.clone(), pts.clone(), metas)
    data = dict(np_sd(m.state_dict()))
    for key in [k_ for k_, v_ in data.items() if v_.size > 200000]:        # stored as their tile: tests/util.py load_golden expands
        w_ = data.pop(key)
        data['sdtile/' + key[3:]] = w_[:41, :37].copy()
        data['sdshape/' + key[3:]] = np.array(w_.shape)
    data.update({'in/pts_feats': pts.numpy(), 'in/img_feats': img.numpy(), 'in/lidar2img': l2i, 'in/input_shape': np.array(shape),
                 'out/pts_feat_conv': pts_conv.numpy(), 'out/new_img_feat': new_img.numpy()})
    for i, t in enumerate(stages):
        data[f'out/stage_{i}'] = t.numpy()
    data['cfg'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'neck_bevfusion_lss.npz'), **data)
    print('neck_bevfusion_lss written;', len(stages), 'stage maps; occupied camera-BEV fraction',
          float((new_img.abs() > 0).float().mean()))


def gen_lss(ref):
    """LiftSplatShoot (necks/lss.py) run from the reference source on CPU (its default voxel_pooling path)."""
    g = torch.Generator().manual_seed(41)
    cfg = dict(img_scale=(32, 64), downsample=4, depth_range=[4.0, 45.0, 1.0], pc_range=[-54, -54, -5, 54, 54, 3],
               grid=6, camC=8, inputC=12, outputC=10)
    with S.cpu_device_patch():
        m = ref.LiftSplatShoot(img_scale=cfg['img_scale'], camera_depth_range=cfg['depth_range'], pc_range=cfg['pc_range'],
                               downsample=4, grid=cfg['grid'], inputC=12, outputC=10, camC=8, newbevpool=False).eval()
    randomize(m, g)
    with torch.no_grad():
        m.frustum.copy_(m.create_frustum())           # randomize() must not touch the (non-trainable) frustum grid
        for prm in m.parameters():                    # the fixed 512-wide BEV encoder: 15-level weights so the fixture
            if prm.numel() > 20000:                   # compresses (2.4 M random floats would be 9 MB)
                prm.copy_(torch.randint(-7, 8, prm.shape, generator=g).float() * 0.005)
    B, N = 2, 3
    x = torch.randn(B, N, 12, 8, 16, generator=g)
    l2i = torch.from_numpy(synthetic_rig(B, N, cfg['img_scale']))
    inv = torch.inverse(l2i)
    rots, trans = inv[..., :3, :3].contiguous(), inv[..., :3, 3].contiguous()
    with torch.no_grad(), S.cpu_device_patch():
        bev, depth = m(x.clone(), rots, trans, img_metas=[{} for _ in range(B)])
    data = dict(np_sd(m.state_dict()))
    data.update({'in/x': x.numpy(), 'in/rots': rots.numpy(), 'in/trans': trans.numpy(), 'in/lidar2img': l2i.numpy(),
                 'out/bev': bev.numpy(), 'out/depth': depth.numpy()})
    data['cfg'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'lss_small.npz'), **data)
    print('lss_small written; occupied BEV fraction', float((bev.abs() > 0).float().mean()))


def gen_get_bboxes_nms(ref):
    """FocalDecoder.get_bboxes with test_cfg.nms_type 'circle' / 'rotate' (FD:1313-1413) executed by the REFERENCE on crafted
    predictions: score fusion, decode + range filter, the per-dataset task tables (FD:1333-1345), task masks, keep indices, the
    200-box cap.  mmdet3d's `circle_nms` / `nms_gpu` (un-vendored) are served by the oracle's restatements (ref_shims.py), so this
    pins the reference's own logic AROUND them.  300 queries of one frame, centres inside an 18 m patch, a third of them jittered
    copies of other queries (rotated IoU > 0.7 pairs); Waymo / rotate keeps more than 200 boxes (the cap)."""
    from oracle import ff3d_oracle as O_

    def craft(seed, K, vel, nus, Nq):
        g = torch.Generator().manual_seed(seed)
        src = torch.randint(0, Nq, (Nq,), generator=g)
        dup = torch.rand(Nq, generator=g) < 0.33                         # jittered copies of query src[q]
        pick = torch.where(dup, src, torch.arange(Nq))
        p_lab = torch.tensor([0.03] * 8 + [0.4, 0.36]) if nus else torch.tensor([0.4, 0.3, 0.3])
        labels = torch.multinomial(p_lab, Nq, replacement=True, generator=g)[pick][None]
        jit = lambda shape, s_: torch.where(dup[None, None], torch.randn(shape, generator=g) * s_, torch.zeros(shape))
        center = (torch.rand(1, 2, Nq, generator=g) * 30 + 60)[:, :, pick] + jit((1, 2, Nq), 0.05)
        preds = dict(heatmap=torch.randn(1, K, Nq, generator=g), query_heatmap_score=torch.rand(1, K, Nq, generator=g),
                     center=center, height=torch.randn(1, 1, Nq, generator=g)[:, :, pick],
                     dim=(torch.rand(1, 3, Nq, generator=g) + 0.5)[:, :, pick] + jit((1, 3, Nq), 0.02),
                     rot=torch.randn(1, 2, Nq, generator=g)[:, :, pick] + jit((1, 2, Nq), 0.02))
        if vel:
            preds['vel'] = torch.randn(1, 2, Nq, generator=g)
        return preds, labels

    def margin(preds, labels, ocfg, K):
        """Smallest distance of a same-task pair from a decision threshold (squared centre distance / rotated IoU vs the task's
        radius; score ties): an fp32 implementation on another device must not be able to flip a comparison."""
        score = preds['heatmap'].sigmoid() * preds['query_heatmap_score'] * F.one_hot(labels, K).permute(0, 2, 1)
        d = O_.bbox_decode(score, preds['rot'].clone(), preds['dim'].clone(), preds['center'].clone(), preds['height'].clone(),
                           preds['vel'].clone() if 'vel' in preds else None, ocfg)[0][0]
        b, sc, l = d['bboxes'], d['scores'], d['labels']
        worst = 100.0 * float((torch.sort(sc).values.diff().abs().min()))      # (score gaps: 1e-6 is clear - same fp32 arithmetic)
        for idx, radius in O_.NMS_TASKS[ocfg.dataset]:
            if radius <= 0:
                continue
            m = torch.zeros_like(sc, dtype=torch.bool)
            for ci in idx:
                m |= l == ci
            xy = b[m][:, :2].numpy().astype(np.float32)
            d2 = ((xy[:, None] - xy[None]) ** 2).sum(-1)
            bev = O_.xywhr2xyxyr(b[m][:, [0, 1, 3, 4, 6]]).numpy()
            iou = O_.boxes_iou_bev(bev, bev)
            off = ~np.eye(len(xy), dtype=bool)
            worst = min(worst, float(np.abs(d2 - radius)[off].min()), float(np.abs(iou - radius)[off].min()))
        return worst

    for dataset, K, vel in (('nuScenes', 10, True), ('Waymo', 3, False)):
        nus = dataset == 'nuScenes'
        heads = dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2))
        if vel:
            heads['vel'] = (2, 2)
        pcr = [-54.0, -54.0] if nus else [-75.2, -75.2]
        vox = 0.075 if nus else 0.1
        pcrange = [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0] if nus else [-80, -80, -10.0, 80, 80, 10.0]
        coder = dict(type='TransFusionBBoxCoder', pc_range=pcr, voxel_size=[vox, vox], out_size_factor=8, post_center_range=pcrange,
                     score_threshold=0.0, code_size=10 if vel else 8)
        C, Nq = 16, 300
        head = ref.FocalDecoder(num_proposals=Nq, hidden_channel=C, num_classes=K, num_decoder_layers=1, num_heads=8,
                                initialize_by_heatmap=True, nms_kernel_size=3, common_heads=heads, bbox_coder=coder,
                                loss_cls=dict(type='FocalLoss', use_sigmoid=True), decoder_cfg=decoder_cfg(C), multiscale=True,
                                bevpos=True, input_img=False, iterbev_wo_img=True,
                                test_cfg=dict(dataset=dataset, grid_size=[1440, 1440, 40], out_size_factor=8, pc_range=pcr,
                                              voxel_size=[vox, vox], nms_type=None, pre_maxsize=60, post_maxsize=40)).eval()
        ocfg = O_.head_config(dataset=dataset, num_classes=K, pc_range=tuple(pcr), voxel_size=(vox, vox), out_size_factor=8,
                              post_center_range=tuple(pcrange), score_threshold=0.0, common_heads=heads)
        for seed in range(61 + K, 161 + K):
            preds, labels = craft(seed, K, vel, nus, Nq)
            mg = margin(preds, labels, ocfg, K)
            if mg >= 1e-4:
                break
        else:
            raise RuntimeError('no seed with clear margins')
        print(dataset, 'seed', seed, 'margin', mg)
        head.query_labels, head.num_proposals = labels, Nq
        data = {'in/' + k_: v_.numpy() for k_, v_ in preds.items()}
        data['in/query_labels'] = labels.numpy()
        counts = {}
        for nms_type in (None, 'circle', 'rotate'):
            head.test_cfg['nms_type'] = nms_type
            with torch.no_grad(), S.cpu_device_patch(), contextlib.redirect_stdout(io.StringIO()):
                boxes, scores, lab = head.get_bboxes([[{k_: v_.clone() for k_, v_ in preds.items()}]],
                                                     [{'box_type_3d': S.LiDARInstance3DBoxes}])[0]
            tag = nms_type or 'none'
            data[f'out/{tag}/bboxes'], data[f'out/{tag}/scores'], data[f'out/{tag}/labels'] = (
                boxes.tensor.numpy(), scores.numpy(), lab.numpy())
            counts[tag] = len(scores)
        cfg = dict(dataset=dataset, num_classes=K, common_heads={a: list(b) for a, b in heads.items()}, pc_range=pcr,
                   voxel_size=[vox, vox], out_size_factor=8, post_center_range=pcrange, score_threshold=0.0, pre_maxsize=60,
                   post_maxsize=40)
        data['cfg'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, f'get_bboxes_nms_{dataset.lower()}.npz'), **data)
        print(f'get_bboxes_nms_{dataset.lower()} written; boxes kept', counts)


def gen_merge_augs(ref):
    """TTA merge (core/post_processing/merge_augs.py:13-184) executed by the REFERENCE function: mapping back through the
    shim box container, per-class rotated NMS + IoU voting (mmdet3d's iou3d ops served by the oracle's restatement of
    iou3d_kernel.cu - those stay "unpinned by execution"), final sort / top-500.  Four augmentation passes over one scene:
    identity, horizontal flip, vertical flip + scale 0.95, both flips + scale 1.05; boxes with velocity (box_dim 9).
    Seeds are searched until every same-class IoU stays 2e-4 clear of the two thresholds (0.1, 0.65), so that an fp32
    implementation on another device cannot flip a comparison."""
    import contextlib
    import io
    import tempfile
    for seed in range(300, 400):
        g = torch.Generator().manual_seed(seed)
        n = 120
        base = torch.zeros(n, 9)
        base[:, :2] = torch.rand(n, 2, generator=g) * 60 - 30
        base[:, 2] = torch.rand(n, generator=g) * 2 - 2
        base[:, 3:6] = torch.rand(n, 3, generator=g) * torch.tensor([1.5, 3.5, 1.0]) + torch.tensor([1.2, 2.5, 1.2])
        base[:, 6] = (torch.rand(n, generator=g) - 0.5) * 6.2
        base[:, 7:] = torch.randn(n, 2, generator=g)
        labels0 = torch.randint(0, 4, (n,), generator=g)
        augs = [(1.0, False, False), (1.0, True, False), (0.95, False, True), (1.05, True, True)]
        results, metas, data = [], [], {}
        for i, (scale, fh, fv) in enumerate(augs):
            keep = torch.rand(n, generator=g) < 0.9
            b = base[keep].clone()
            b[:, :2] += torch.randn(b.shape[0], 2, generator=g) * 0.06
            b[:, 3:6] *= 1 + torch.randn(b.shape[0], 3, generator=g) * 0.02
            b[:, 6] += torch.randn(b.shape[0], generator=g) * 0.03
            fwd = S.LiDARInstance3DBoxes(b.clone(), box_dim=9)          # forward transform = what the detector saw
            fwd.scale(scale)
            if fv:
                fwd.flip('vertical')
            if fh:
                fwd.flip('horizontal')
            sc = torch.rand(b.shape[0], generator=g)
            lb = labels0[keep]
            results.append(dict(boxes_3d=fwd, scores_3d=sc, labels_3d=lb))
            metas.append([dict(pcd_scale_factor=scale, pcd_horizontal_flip=fh, pcd_vertical_flip=fv, sample_idx=0)])
            data[f'in/boxes_{i}'], data[f'in/scores_{i}'], data[f'in/labels_{i}'] = fwd.tensor.numpy().copy(), sc.numpy(), lb.numpy()
            data[f'in/aug_{i}'] = np.asarray([scale, float(fh), float(fv)], dtype=np.float32)
        # IoU margins of the mapped-back set (same-class pairs only are ever compared)
        rb = torch.cat([S.shim_bbox3d_mapping_back(r['boxes_3d'], m[0]['pcd_scale_factor'], m[0]['pcd_horizontal_flip'],
                                                   m[0]['pcd_vertical_flip']).tensor for r, m in zip(results, metas)])
        rl = torch.cat([r['labels_3d'] for r in results])
        bev = S.shim_xywhr2xyxyr(rb[:, [0, 1, 3, 4, 6]]).numpy()
        from oracle import ff3d_oracle as O
        iou = torch.from_numpy(O.boxes_iou_bev(bev, bev))
        same = rl[:, None] == rl[None, :]
        if ((iou[same] - 0.1).abs() < 2e-4).any() or ((iou[same] - 0.65).abs() < 2e-4).any():
            continue
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
            os.chdir(tmp)                                   # the reference pickles its inputs into ./merge_augs_initial_results/
            try:
                out = ref.merge_augs.merge_aug_bboxes_3d(results, metas, S.AttrDict(use_rotate_nms=True, nms_thr=0.01, max_num=83))
            finally:
                os.chdir(cwd)
        data['out/boxes'], data['out/scores'] = out['boxes_3d'].tensor.numpy(), out['scores_3d'].numpy()
        data['out/labels'] = out['labels_3d'].numpy()
        data['cfg'] = np.frombuffer(json.dumps(dict(seed=seed, n_aug=len(augs))).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, 'merge_augs.npz'), **data)
        print('merge_augs written; seed', seed, 'in', len(rb), 'out', len(out['scores_3d']))
        return
    raise RuntimeError('no seed with an IoU margin')


def gen_train(ref):
    """Training targets + losses of the head (FD:994-1311) executed by the REFERENCE: its HungarianAssigner3D and match
    costs (core/bbox/assigners/hungarian_assigner.py), FocalDecoder.get_targets / get_targets_single / loss and its bbox
    coder run unmodified on the predictions of its own (eval-mode) forward; the mmdet / mmdet3d pieces they call
    (FocalLossCost, BboxOverlaps3D, AssignResult, PseudoSampler, losses, Gaussian drawing) are the restatements of
    oracle/train_oracle.py.  add_gt_groups = 0: the ground-truth query groups exist only in the training-mode forward,
    which is not mirrored."""
    g = torch.Generator().manual_seed(41)
    C, K, Hb, k, D, B = 32, 10, 36, 20, 2, 2
    pcr = [-54.0, -54.0]
    vox = 2 * abs(pcr[0]) / (Hb * 8)
    heads = dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2))
    coder = dict(type='TransFusionBBoxCoder', pc_range=pcr, voxel_size=[vox, vox], out_size_factor=8,
                 post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.0, code_size=10)
    train_cfg = dict(dataset='nuScenes',
                     assigner=dict(type='HungarianAssigner3D', iou_calculator=dict(type='BboxOverlaps3D', coordinate='lidar'),
                                   cls_cost=dict(type='FocalLossCost', gamma=2, alpha=0.25, weight=0.15),
                                   reg_cost=dict(type='BBoxBEVL1Cost', weight=0.25), iou_cost=dict(type='IoU3DCost', weight=0.25)),
                     pos_weight=-1, gaussian_overlap=0.1, min_radius=2, grid_size=[Hb * 8, Hb * 8, 40],
                     voxel_size=[vox, vox, 0.2], out_size_factor=8, code_weights=[1.0] * 8 + [0.2, 0.2],
                     point_cloud_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0])
    loss_cfgs = dict(loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2, alpha=0.25, reduction='mean', loss_weight=1.0),
                     loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=0.25),
                     loss_heatmap=dict(type='GaussianFocalLoss', reduction='mean', loss_weight=1.0))

    def attr(d):
        return S.AttrDict({a: attr(b) if isinstance(b, dict) else b for a, b in d.items()})
    kw = dict(reuse_first_heatmap=True, extra_feat=True, roi_feats=7, roi_dropout_rate=0.1, roi_based_reg=True,
              roi_expand_ratio=1.2, hidden_channel_roi=48, multiscale=True, multistage_heatmap=2, mask_heatmap_mode='poscls',
              input_img=False, iterbev_wo_img=True, bevpos=True, num_proposals=k, hidden_channel=C, num_classes=K,
              num_decoder_layers=D, num_heads=8, initialize_by_heatmap=True, nms_kernel_size=3, common_heads=heads,
              bbox_coder=coder, decoder_cfg=decoder_cfg(C), add_gt_groups=0, gt_center_limit=5,
              train_cfg=attr(train_cfg), test_cfg=dict(dataset='nuScenes', grid_size=[Hb * 8, Hb * 8, 40], out_size_factor=8,
                                                         pc_range=pcr, voxel_size=[vox, vox], nms_type=None), **loss_cfgs)
    head = ref.FocalDecoder(**kw).eval()
    randomize(head, g)
    f0 = torch.randn(B, C, Hb, Hb, generator=g)
    maps = [torch.randn(B, C, Hb, Hb, generator=g) for _ in range(3)]
    with torch.no_grad(), S.cpu_device_patch():
        preds = head([f0.clone(), [m.clone() for m in maps]], None, [{}] * B)
    gts, labels = [], []
    for b in range(B):
        n = 9 + 4 * b
        t = torch.zeros(n, 9)
        t[:, :2] = torch.rand(n, 2, generator=g) * 90 - 45
        t[:, 2] = torch.rand(n, generator=g) * 2 - 2.5
        t[:, 3:6] = torch.rand(n, 3, generator=g) * torch.tensor([2.0, 4.0, 1.5]) + torch.tensor([0.6, 0.8, 1.0])
        t[:, 6] = (torch.rand(n, generator=g) - 0.5) * 6.2
        t[:, 7:] = torch.randn(n, 2, generator=g)
        gts.append(S.LiDARInstance3DBoxes(t, box_dim=9))
        labels.append(torch.randint(0, K, (n,), generator=g))
    # move a few ground-truth boxes onto predicted boxes so that the IoU / centre-limit terms are exercised
    p0 = preds[0][0]
    with torch.no_grad(), S.cpu_device_patch():
        dec = head.bbox_coder.decode(*(copy.deepcopy(p0[q]) for q in ('heatmap', 'rot', 'dim', 'center', 'height', 'vel')))
    for b in range(B):
        pick = torch.randperm(k * 3 * D, generator=g)[:4]
        bx = dec[b]['bboxes'][pick].clone()
        gts[b].tensor[:4, :7] = bx[:, :7] + torch.randn(4, 7, generator=g) * 0.05
        gts[b].tensor[:4, 3:6] = bx[:, 3:6].clamp(0.3, 8.0)
    with torch.no_grad(), S.cpu_device_patch():
        targets = head.get_targets(gts, labels, preds[0])
        losses = head.loss(gts, labels, [[dict(p0)]])
    data = dict(np_sd(head.state_dict()))
    data['in/pts_feat_conv'] = f0.numpy()
    for i, m in enumerate(maps):
        data[f'in/stage_{i}'] = m.numpy()
    for b in range(B):
        data[f'in/gt_boxes_{b}'], data[f'in/gt_labels_{b}'] = gts[b].tensor.numpy(), labels[b].numpy()
    for key, v in p0.items():
        if torch.is_tensor(v):
            data['pred/' + key] = v.numpy()
        else:
            for i, t in enumerate(v):
                data[f'pred/{key}/{i}'] = t.numpy()
    names = ('labels', 'label_weights', 'bbox_targets', 'bbox_weights', 'ious', 'num_pos', 'matched_ious', 'heatmap')
    for nme, v in zip(names, targets):
        data['out/' + nme] = v.numpy() if torch.is_tensor(v) else np.asarray(v)
    for nme, v in losses.items():
        data['loss/' + nme] = np.asarray(float(v))
    cfg = dict(head=dict(num_proposals=k, hidden_channel=C, num_classes=K, num_decoder_layers=D, grid=Hb, multistage_heatmap=2,
                         reuse_first_heatmap=True, extra_feat=True, roi_feats=7, roi_expand_ratio=1.2, roi_based_reg=True,
                         hidden_channel_roi=48, ffn_channels=64, common_heads={a: list(b) for a, b in heads.items()},
                         pc_range=pcr, voxel_size=[vox, vox], out_size_factor=8, post_center_range=coder['post_center_range'],
                         score_threshold=0.0, gt_center_limit=5, nms_kernel_size=3, dataset='nuScenes'),
               train_cfg=train_cfg, losses=loss_cfgs)
    data['cfg'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'train_targets.npz'), **data)
    print('train_targets written; num_pos', int(targets[5]), 'matched_ious %.4f' % float(targets[6]),
          {a: round(float(b), 5) for a, b in losses.items()})


def gen_heuristic_assigner(ref):
    """HeuristicAssigner3D (hungarian_assigner.py:49-91) executed by the REFERENCE class on three cases: class-aware with a
    tight radius (some boxes unassigned, several boxes claiming one proposal), class-agnostic, and a radius nobody meets."""
    g = torch.Generator().manual_seed(61)
    data = {}
    for i, (P, G, thre, aware) in enumerate([(60, 25, 6.0, True), (40, 30, 100.0, False), (30, 8, 0.05, True)]):
        def boxes(n):
            t = torch.zeros(n, 9)
            t[:, :2] = torch.rand(n, 2, generator=g) * 40 - 20
            t[:, 2] = torch.rand(n, generator=g) * 2 - 2.5
            t[:, 3:6] = torch.rand(n, 3, generator=g) * torch.tensor([2.0, 4.0, 1.5]) + torch.tensor([0.6, 0.8, 1.0])
            t[:, 6] = torch.rand(n, generator=g) * 6.28 - 3.14
            return t
        pred, gt = boxes(P), boxes(G)
        pred[:G // 2, :7] = gt[:G // 2, :7] + torch.randn(G // 2, 7, generator=g) * 0.3      # some real overlaps
        pred[G // 2, :2] = gt[0, :2] + 0.01                                                 # two boxes claiming one proposal
        gt[1, :2] = gt[0, :2] + 0.5
        gl, ql = torch.randint(0, 3, (G,), generator=g), torch.randint(0, 3, (P,), generator=g)
        asg = ref.hungarian_assigner.HeuristicAssigner3D(dist_thre=thre, iou_calculator=dict(type='BboxOverlaps3D', coordinate='lidar'))
        res = asg.assign(pred.clone(), gt.clone(), None, gl.clone(), ql.clone() if aware else None)
        data.update({f'c{i}/pred': pred.numpy(), f'c{i}/gt': gt.numpy(), f'c{i}/gt_labels': gl.numpy(), f'c{i}/query_labels': ql.numpy(),
                     f'c{i}/dist_thre': np.float32(thre), f'c{i}/aware': np.int32(aware), f'c{i}/gt_inds': res.gt_inds.numpy(),
                     f'c{i}/max_overlaps': res.max_overlaps.numpy(), f'c{i}/labels': res.labels.numpy()})
        print('heuristic assigner case', i, 'assigned', int((res.gt_inds > 0).sum()), 'of', G, 'boxes')
    np.savez_compressed(os.path.join(OUT, 'heuristic_assigner.npz'), **data)


def gen_train_step(ref, name, seed, waymo):
    """One training step of the head executed by the REFERENCE in train() mode: FocalDecoder.forward with ground truth
    (batch-statistics BatchNorm; with ``add_gt_groups`` the noised ground-truth query groups FD:377-520 and their attention masks
    FD:849-858), .loss, and backward of the summed loss terms - predictions, losses, the gradient of every parameter and of every
    input map, and the BatchNorm buffers after the step.  The un-vendored decoder is the oracle's restatement behind the shim (it
    has no dropout, i.e. the decoder's dropout probabilities are 0 here; roi_dropout_rate = 0); the torch.rand draws of the
    ground-truth groups are recorded for replay."""
    g = torch.Generator().manual_seed(seed)
    C, Hb, k, D, B = 32, 24, 16, 2, 2
    K = 3 if waymo else 10
    dataset = 'Waymo' if waymo else 'nuScenes'
    lim = 75.2 if waymo else 54.0
    pcr = [-lim, -lim]
    vox = 2 * lim / (Hb * 8)
    heads = dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2))
    if not waymo:
        heads['vel'] = (2, 2)
    code = 8 if waymo else 10
    coder = dict(type='TransFusionBBoxCoder', pc_range=pcr, voxel_size=[vox, vox], out_size_factor=8,
                 post_center_range=[-lim - 5, -lim - 5, -10.0, lim + 5, lim + 5, 10.0], score_threshold=0.0, code_size=code)
    full_range = [-lim, -lim, -2.0 if waymo else -5.0, lim, lim, 4.0 if waymo else 3.0]
    train_cfg = dict(dataset=dataset,
                     assigner=dict(type='HungarianAssigner3D', iou_calculator=dict(type='BboxOverlaps3D', coordinate='lidar'),
                                   cls_cost=dict(type='FocalLossCost', gamma=2, alpha=0.25, weight=0.6 if waymo else 0.15),
                                   reg_cost=dict(type='BBoxBEVL1Cost', weight=2.0 if waymo else 0.25),
                                   iou_cost=dict(type='IoU3DCost', weight=2.0 if waymo else 0.25)),
                     pos_weight=-1, gaussian_overlap=0.1, min_radius=2, grid_size=[Hb * 8, Hb * 8, 40],
                     voxel_size=[vox, vox, 0.2], out_size_factor=8, code_weights=[1.0] * 8 + ([] if waymo else [0.2, 0.2]),
                     point_cloud_range=full_range)
    loss_cfgs = dict(loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2, alpha=0.25, reduction='mean', loss_weight=1.0),
                     loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=2.0 if waymo else 0.25),
                     loss_heatmap=dict(type='GaussianFocalLoss', reduction='mean', loss_weight=1.0))

    def attr(d):
        return S.AttrDict({a: attr(b) if isinstance(b, dict) else b for a, b in d.items()})
    extra_kw = dict(add_gt_groups=3, add_gt_groups_noise='box,1', add_gt_groups_noise_box='gtnoise', add_gt_pos_thresh=5.,
                    add_gt_pos_boxnoise_thresh=0.75) if waymo else dict(add_gt_groups=0)
    kw = dict(reuse_first_heatmap=True, extra_feat=True, roi_feats=7, roi_dropout_rate=0.0, roi_based_reg=True,
              roi_expand_ratio=1.2, hidden_channel_roi=16, multiscale=True, multistage_heatmap=2, mask_heatmap_mode='poscls',
              input_img=False, iterbev_wo_img=True, bevpos=True, num_proposals=k, hidden_channel=C, num_classes=K,
              num_decoder_layers=D, num_heads=8, initialize_by_heatmap=True, nms_kernel_size=3, common_heads=heads,
              bbox_coder=coder, decoder_cfg=decoder_cfg(C), gt_center_limit=5, bn_momentum=0.1,
              train_cfg=attr(train_cfg), test_cfg=dict(dataset=dataset, grid_size=[Hb * 8, Hb * 8, 40], out_size_factor=8,
                                                         pc_range=pcr, voxel_size=[vox, vox], nms_type=None),
              **extra_kw, **loss_cfgs)
    head = ref.FocalDecoder(**kw).train()
    randomize(head, g)
    f0 = torch.randn(B, C, Hb, Hb, generator=g)
    maps = [torch.randn(B, C, Hb, Hb, generator=g) for _ in range(3)]
    gts, labels = [], []
    for b in range(B):
        n = 7 + 3 * b
        t = torch.zeros(n, code - 1)
        t[:, :2] = (torch.rand(n, 2, generator=g) * 1.6 - 0.8) * lim
        t[:, 2] = torch.rand(n, generator=g) * 2 - 2.5
        t[:, 3:6] = torch.rand(n, 3, generator=g) * torch.tensor([2.0, 4.0, 1.5]) + torch.tensor([0.6, 0.8, 1.0])
        t[:, 6] = (torch.rand(n, generator=g) - 0.5) * 6.2
        if code == 10:
            t[:, 7:] = torch.randn(n, 2, generator=g)
        gts.append(S.LiDARInstance3DBoxes(t, box_dim=code - 1))
        labels.append(torch.randint(0, K, (n,), generator=g))
    # a dry run on a copy (same train-mode arithmetic) to move a few ground-truth boxes onto predicted boxes, so that the IoU
    # cost, the centre limit and positive regression targets are exercised
    dry = copy.deepcopy(head)
    with torch.no_grad(), S.cpu_device_patch():
        torch.manual_seed(seed)
        p0 = dry([f0.clone(), [m.clone() for m in maps]], None, [{}] * B, gt_bboxes_3d=gts, gt_labels_3d=labels)[0][0]
        dec = dry.bbox_coder.decode(*(copy.deepcopy(p0[q]) for q in ('heatmap', 'rot', 'dim', 'center', 'height')),
                                    copy.deepcopy(p0['vel']) if 'vel' in p0 else None)
    for b in range(B):
        pick = torch.randperm(p0['center'].shape[-1], generator=g)[:4]
        bx = dec[b]['bboxes'][pick].clone()
        gts[b].tensor[:4, :7] = bx[:, :7] + torch.randn(4, 7, generator=g) * 0.05
        # (sizes: a relative offset - a target that equals the prediction to rounding would make the sign of the L1 gradient noise)
        gts[b].tensor[:4, 3:6] = (bx[:, 3:6] * (1 + 0.05 * (torch.rand(4, 3, generator=g) + 0.2))).clamp(0.3, 8.0)
    ins = [f0.clone().requires_grad_(True)] + [m.clone().requires_grad_(True) for m in maps]
    sd0 = {a: b.clone() for a, b in head.state_dict().items()}          # parameters and BatchNorm buffers before the step
    rands = []
    # the step is evaluated at de-singularised sampling locations (oracle.ff3d_oracle.desingularise_sampling: the location
    # gradient of bilinear sampling is discontinuous at pixel crossings); the moved coordinates are part of the fixture
    from oracle import ff3d_oracle as O
    msda_fix = []

    def loc_hook(loc, shapes):
        new, idx, val = O.desingularise_sampling(loc, shapes)
        msda_fix.append((idx, val))
        return new
    O.MSDA_LOC_HOOK = loc_hook
    try:
        with S.cpu_device_patch(rand_log=rands):
            torch.manual_seed(seed + 1)
            preds = head([ins[0], list(ins[1:])], None, [{}] * B, gt_bboxes_3d=gts, gt_labels_3d=labels)
            p0 = dict(preds[0][0])
            dense_list = list(p0['dense_heatmap'])                 # .loss concatenates the list in place
            losses = head.loss(gts, labels, preds)
    finally:
        O.MSDA_LOC_HOOK = None
    total = sum(v for n_, v in losses.items() if 'loss' in n_)
    total.backward()
    # The L1 terms have a sign() in their gradient: a regression element whose prediction sits within rounding of its target
    # would make the golden gradient a coin toss on another machine.  Measure the smallest |prediction - target| that carries
    # weight (targets recomputed by the reference's own get_targets on the same predictions) and insist on a margin.
    with torch.no_grad(), S.cpu_device_patch():
        tg = head.get_targets(gts, labels, [dict(p0, dense_heatmap=dense_list)])
    bt, bw = tg[2], tg[3]
    keys = ['center', 'height', 'dim', 'rot'] + (['vel'] if 'vel' in p0 else [])
    pred_all = torch.cat([p0[q] for q in keys], 1).permute(0, 2, 1).detach()
    margin = float((pred_all - bt).abs()[bw > 0].min())
    if 'center_gtgroups' in p0:
        pq = torch.cat([p0[q + '_gtgroups'] for q in ['center', 'height', 'rot', 'dim'] + (['vel'] if 'vel' in p0 else [])], 1)
        pq = pq.permute(0, 2, 1).detach()
        tq = torch.zeros(pq.shape[0], head.max_num_gts, head.bbox_coder.code_size)
        for b_, g_ in enumerate(gts):
            tq[b_, :len(g_.tensor)] = head.bbox_coder.encode(g_.tensor)
        tq = tq.repeat(1, head.add_gt_groups * head.num_decoder_layers, 1)
        wq = (p0['batch_valid_gt_mask'].float()[:, :, None].repeat(1, head.num_decoder_layers, 1)
              * (p0['batch_gt_query_labels'].repeat(1, head.num_decoder_layers) != head.num_classes)[..., None].float())
        margin = min(margin, float((pq - tq).abs()[wq.expand_as(pq) > 0].min()))
    print(name, 'smallest weighted |prediction - target| of an L1 term: %.3e' % margin)
    assert margin > 5e-5, 'an L1 term sits on its kink: change the seed'
    data = dict(np_sd(sd0))
    for a, b in head.state_dict().items():
        if 'running_' in a:
            data['bn_after/' + a] = b.numpy().copy()
    data['in/pts_feat_conv'] = f0.numpy()
    for i, m in enumerate(maps):
        data[f'in/stage_{i}'] = m.numpy()
    for b in range(B):
        data[f'in/gt_boxes_{b}'], data[f'in/gt_labels_{b}'] = gts[b].tensor.numpy(), labels[b].numpy()
    for i, r in enumerate(rands):
        data[f'rand/{i}'] = r.numpy()
    for i, (idx, val) in enumerate(msda_fix):              # one entry per deformable-attention call, in call order
        data[f'msda_fix/{i}/idx'], data[f'msda_fix/{i}/val'] = idx.numpy(), val.numpy()
    print(name, 'de-singularised sampling coordinates per MSDA call:', [len(i_) for i_, _ in msda_fix])
    p0['dense_heatmap'] = dense_list
    for key, v in p0.items():
        if torch.is_tensor(v):
            data['pred/' + key] = v.detach().numpy()
        else:
            for i, t in enumerate(v):
                data[f'pred/{key}/{i}'] = t.detach().numpy()
    for nme, v in losses.items():
        data['loss/' + nme] = np.asarray(float(v))
    for nme, p in head.named_parameters():
        data['grad/' + nme] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        data['hasgrad/' + nme] = np.asarray(p.grad is not None)
    for i, t in enumerate(ins):
        data[f'gin/{i}'] = t.grad.numpy()
    cfg = dict(head=dict(num_proposals=k, hidden_channel=C, num_classes=K, num_decoder_layers=D, grid=Hb, multistage_heatmap=2,
                         reuse_first_heatmap=True, extra_feat=True, roi_feats=7, roi_expand_ratio=1.2, roi_based_reg=True,
                         hidden_channel_roi=16, ffn_channels=64, common_heads={a: list(b) for a, b in heads.items()},
                         pc_range=pcr, voxel_size=[vox, vox], out_size_factor=8, post_center_range=coder['post_center_range'],
                         score_threshold=0.0, gt_center_limit=5, nms_kernel_size=3, dataset=dataset, code_size=code, **extra_kw),
               train_cfg=train_cfg, losses=loss_cfgs)
    data['cfg'] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **data)
    print(name, 'written;', {a: round(float(b), 5) for a, b in losses.items()}, 'rand draws', len(rands),
          'params without grad', [n_ for n_, p in head.named_parameters() if p.grad is None])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    only = sys.argv[sys.argv.index('--only') + 1] if '--only' in sys.argv else None
    if only == 'heuristic_assigner':           # python -m oracle.gen_golden --only heuristic_assigner
        gen_heuristic_assigner(S.load_reference())
        return
    if only == 'get_bboxes_nms':               # python -m oracle.gen_golden --only get_bboxes_nms
        gen_get_bboxes_nms(S.load_reference())
        return
    if only == 'neck_lss':                     # python -m oracle.gen_golden --only neck_lss
        gen_neck_lss(S.load_reference())
        return
    if only == 'coder_threshold':              # python -m oracle.gen_golden --only coder_threshold
        gen_coder_threshold(S.load_reference())
        return
    if only == 'head_options':                 # python -m oracle.gen_golden --only head_options
        gen_head_options(S.load_reference())
        return
    if only == 'head_heatbox':                 # python -m oracle.gen_golden --only head_heatbox
        gen_head_heatbox(S.load_reference())
        return
    if only == 'train_step':                   # python -m oracle.gen_golden --only train_step
        ref = S.load_reference()
        gen_train_step(ref, 'train_step_nus', 51, waymo=False)
        gen_train_step(ref, 'train_step_waymo', 52, waymo=True)
        return
    if only == 'decoder_hf':                   # python -m oracle.gen_golden --only decoder_hf
        gen_decoder_hf()
        return
    gen_msda_hf()              # HF transformers first: it must see the real (absent) torchvision, not the shim's stand-in
    gen_decoder_hf()
    ref = S.load_reference()
    gen_posembed(ref)
    gen_coder(ref)
    gen_i2p(ref)
    gen_lss(ref)
    gen_merge_augs(ref)
    gen_get_bboxes_nms(ref)
    gen_train(ref)
    gen_heuristic_assigner(ref)
    gen_train_step(ref, 'train_step_nus', 51, waymo=False)
    gen_train_step(ref, 'train_step_waymo', 52, waymo=True)
    gen_neck(ref, 'neck_mb2_lidar', 31, 'bevfusionmb2', with_img=False)      # FocalFormer3D_L-like neck
    gen_neck(ref, 'neck_bevfusion_cam', 32, 'bevfusion', with_img=True)      # FocalFormer3D_LC_Proj-like neck
    gen_neck_lss(ref)                                                        # FocalFormer3D_LC-like neck (LSS camera branch)
    # FocalFormer3D_L-like: reuse_first_heatmap, 2+1 stages, RoI 7x7, 2 decoder stages
    gen_head(ref, 'head_focal_L', 21, C=32, K=10, Hb=36, k=20, dataset='nuScenes', multistage=2, reuse=True,
             extra=True, roi=7, D=2)
    # FocalFormer3D_LC-like: no reuse (first stage heatmap from stage map 0), input_img=True
    gen_head(ref, 'head_focal_LC', 22, C=16, K=10, Hb=28, k=12, dataset='nuScenes', multistage=2, reuse=False,
             extra=True, roi=7, D=2, input_img=True, iterbev_wo_img=False)
    # DeformFormer3D_L-like: single-stage branch, two-heatmap mean, argsort top-k, no RoI, D=1
    gen_head(ref, 'head_deform_L', 23, C=16, K=10, Hb=28, k=30, dataset='nuScenes', multistage=None, reuse=False,
             extra=False, roi=0, D=1)
    # Waymo-like: K=3 (small classes 1,2), no velocity head, code_size 8
    gen_head(ref, 'head_waymo', 24, C=16, K=3, Hb=32, k=16, dataset='Waymo', multistage=2, reuse=True,
             extra=True, roi=7, D=2, vel=False)
    gen_head_options(ref)
    gen_head_heatbox(ref)


def gen_head_options(ref):
    """Inference-path options of FocalDecoder the four config-shaped fixtures leave at their defaults."""
    # FocalFormer3D_Waymo15_L.py:197,227: the 14 x 14 RoI grid + class-aware regression - one regression column block per
    # class, picked by the query's label (FD:940-943)
    gen_head(ref, 'head_opt_classaware', 25, C=16, K=3, Hb=24, k=8, dataset='Waymo', multistage=2, reuse=True,
             extra=True, roi=14, D=2, vel=False, classaware=True)
    # mask_heatmap_mode='pos': the positive mask blanks the selected cell of the selected class only (FD:725-728)
    gen_head(ref, 'head_opt_posmask', 26, C=16, K=10, Hb=24, k=12, dataset='nuScenes', multistage=2, reuse=True,
             extra=True, roi=0, D=1, mask_mode='pos')
    # DeformFormer3D_Waymo_L.py / DeformFormer3D_Waymo15_L.py: the single-stage branch WITHOUT the second heatmap
    # (input_img=False, iterbev_wo_img=False: FD:550-552 - heatmap = sigmoid(dense_heatmap), queries gathered from pts_inputs[0])
    gen_head(ref, 'head_opt_singleheat', 28, C=16, K=3, Hb=24, k=20, dataset='Waymo', multistage=None, reuse=False,
             extra=False, roi=0, D=1, vel=False, input_img=False, iterbev_wo_img=False)
    # multiscale=False, bevpos=False: one BEV level in the value, no position embedding on it (FD:835-838, 887-888)
    gen_head(ref, 'head_opt_singlescale', 27, C=16, K=10, Hb=24, k=12, dataset='nuScenes', multistage=2, reuse=True,
             extra=True, roi=0, D=1, multiscale=False, bevpos=False)


def gen_head_heatbox(ref):
    """The heatmap_box branch of the inference path (no shipped config enables it)."""
    # heatmap_box + thin_heatmap_box (FD:231-287, 606-660, 708-722): a (conv, conv) task head per stage regresses a box per cell and
    # task; the queries start from those boxes (RoI features already at the first decoder stage, FD:890)
    gen_head(ref, 'head_opt_heatbox', 29, C=16, K=10, Hb=24, k=12, dataset='nuScenes', multistage=2, reuse=True,
             extra=True, roi=7, D=2, heatmap_box=True)
    # ... + mask_heatmap_mode='boxcls' (FD:732-770): cells inside a selected query's box are masked for that query's class
    # (mmdet3d's points_in_boxes_gpu: served by the oracle's restatement)
    gen_head(ref, 'head_opt_boxcls', 30, C=16, K=10, Hb=24, k=12, dataset='nuScenes', multistage=2, reuse=True,
             extra=True, roi=0, D=1, heatmap_box=True, mask_mode='boxcls', weight_gain={'multi_stage_task_heads': 1.6})


if __name__ == '__main__':
    main()
