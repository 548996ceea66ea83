"""Training-mode forward of the head: ``FocalDecoder.forward`` with ``self.training`` (FD:522-992) - SURVEY.md §8f rank 4.

Same algorithm as the inference path, but every learnable layer has to be differentiable, so the roles are split the other way
round than in ``FocalDecoder._forward_eval``:

* index work stays on the hand-written kernels, exactly as at inference: sigmoid * mask + local-max NMS (``ff3d_heatmap_nms``),
  the deterministic top-k (``ff3d_topk``), the accumulated-mask update and the per-query positions / scores / labels
  (``ff3d_query_gather``), the sine embeddings (``ff3d_sine_embed``: their inputs are detached positions, FD:869/947);
* the deformable gather is ``MultiScaleDeformableAttnFunction`` (``ff3d_msda_fwd`` / ``ff3d_msda_bwd``);
* the RoI feature read is ``RoIGridSampleFunction`` (``ff3d_roi_grid_sample`` / ``ff3d_roi_grid_sample_bwd`` on the channels-last
  pyramid; the framework's grid_sample backward was 30 % of a training step);
* convolutions, BatchNorms (batch statistics, ``bn_momentum``), linears, LayerNorms, dropouts and the query-feature ``gather``
  that carries gradient to the BEV maps are the framework's autograd ops on the module's own parameters.

Outputs: the reference's dict (``dense_heatmap`` logits with gradient, ``multistage_masks``, per-stage predictions, and with
``add_gt_groups > 0`` the ``*_gtgroups`` predictions, ``batch_valid_gt_mask``, ``batch_gt_query_labels``), consumed unchanged by
``FocalDecoder.loss``.  ``heatmap_box`` (inference-only here: forward_train raises) / ``boxpos`` (the constructor rejects it)
branches: not built for training - no shipped config enables them.  The reference draws the ground-truth-group noise with ``torch.rand(..., device='cuda')`` (FD:408,502): here through
``head._rand(shape, device)``, so a test can replay a recorded draw.
"""
import torch
import torch.nn.functional as F

from . import ops
from .layers import gen_sineembed_for_position


def _box_tensor(b):
    return b.tensor if hasattr(b, 'tensor') else b


def _static(head, key, make):
    """Device tensors that depend on the configuration only (grids, ranges, their sine embeddings): built once per head and device.
    Every ``torch.tensor(list, device=...)`` / ``cpu_tensor.to(device)`` is a blocking host-to-device copy - two dozen of them per
    step were 7 ms of a host-bound 18 ms forward (round 6, tools/profile_host_train.py)."""
    cache = head.__dict__.setdefault('_train_static', {})
    if key not in cache:
        with torch.no_grad():
            cache[key] = make()
    return cache[key]


_UNIT_CORNERS = {}


def bev_corners(boxes):
    """(n, >=7) LiDAR boxes -> (n, 4, 2): the four BEV corners in the order the reference reads them out of mmdet3d 0.17.1
    ``LiDARInstance3DBoxes.corners`` (un-vendored, Appendix A.5) at FD:397: ``corners.reshape(-1,4,2,3)[:, :4, 0, :2]`` =
    normalised (-.5,-.5), (-.5,.5), (.5,-.5), (.5,.5) times (x_size, y_size), rotated by yaw (``rotation_3d_in_axis`` axis 2:
    x' = x cos + y sin, y' = -x sin + y cos), translated to the centre."""
    key = (boxes.device, boxes.dtype)
    if key not in _UNIT_CORNERS:
        _UNIT_CORNERS[key] = boxes.new_tensor([[-0.5, -0.5], [-0.5, 0.5], [0.5, -0.5], [0.5, 0.5]])
    p = _UNIT_CORNERS[key][None] * boxes[:, None, 3:5]
    s, c = torch.sin(boxes[:, 6])[:, None], torch.cos(boxes[:, 6])[:, None]
    x, y = p[..., 0], p[..., 1]
    return torch.stack([x * c + y * s, -x * s + y * c], -1) + boxes[:, None, :2]


def generate_gt_groups(head, query_feat, query_pos, query_heatmap_score, lidar_feat, lidar_feat_flatten, bev_pos, heatmap,
                       gt_bboxes_3d, gt_labels_3d, dense_heatmap_boxes=None, query_box=None):
    """FD:377-520: ``add_gt_groups`` noised copies of every ground-truth centre become extra queries (feature gathered at the
    noised BEV cell + class embedding; label = the gt class, or background when the noise moved the centre too far).
    query_feat (B,C,Nq), query_pos (B,Nq,2), query_heatmap_score (B,K,Nq), lidar_feat_flatten (B,C,HW), bev_pos (B|1,HW,2),
    heatmap (B,K,HW) -> (query_feat, query_pos, query_heatmap_score, batch_valid_gt_mask (B,G) bool, batch_gt_query_labels
    (B,G) int64) with G = max_num_gts * add_gt_groups appended queries (invalid slots zeroed)."""
    if dense_heatmap_boxes is not None or query_box is not None:
        raise NotImplementedError('generate_gt_groups with dense_heatmap_boxes needs the heatmap_box branch (FD:489-516)')
    dev = query_pos.device
    B, K, G, M = len(gt_bboxes_3d), head.num_classes, head.add_gt_groups, head.max_num_gts
    pcr = _static(head, ('pcr', dev), lambda: torch.as_tensor(head.train_cfg['point_cloud_range'], dtype=torch.float32, device=dev))
    rows, cols = lidar_feat.shape[-2:]                          # the reference's (W, H) = (y, x) extents, FD:446
    spec = head.add_gt_groups_noise.split(',')
    kind = spec[0]
    valid = torch.zeros(B, M * G, dtype=torch.bool, device=dev)
    cells, labels_out = [], []
    for b in range(B):
        box = _box_tensor(gt_bboxes_3d[b]).to(dev).float()
        n = box.shape[0]
        for g in range(G):
            valid[b, g * M:g * M + n] = True
        pad = M - n
        centre = F.pad(box[:, :2], (0, 0, 0, pad)).repeat(G, 1)
        lab = F.pad(gt_labels_3d[b].to(dev), (0, pad), value=K).repeat(G)          # padding = background class
        corners = F.pad(bev_corners(box), (0, 0, 0, 0, 0, pad)).repeat(G, 1, 1)
        meta = head._rand((M * G, 2), dev) * 2 - 1                                 # FD:408
        if kind.startswith('rect'):                                                # axis-aligned extent of the box
            ext = corners.max(1)[0] - corners.min(1)[0]
            shift = ext / 2. * (float(spec[1]) * meta)
            centre = centre + shift
        elif kind.startswith('cam'):                                               # extent in the radial / tangential frame
            u = centre / (centre.norm(dim=1) + 1e-6)[:, None]
            frame = torch.stack([u, torch.stack([u[:, 1], -u[:, 0]], 1)], -1)
            t = corners.matmul(frame)
            ext = t.max(1)[0] - t.min(1)[0]
            scale = (torch.as_tensor([float(v) for v in spec[1].split('-')], device=dev)[None] if '-' in spec[1]
                     else float(spec[1]))
            shift = ext / 2. * (scale * meta)
            centre = centre + shift[:, None].matmul(frame)[:, 0]
        elif kind.startswith('box'):                                               # along the box's own two axes
            noise = float(spec[1]) * meta
            shift = (corners[:, 2] - corners[:, 0]) / 2. * noise[:, 0:1] + (corners[:, 1] - corners[:, 0]) / 2. * noise[:, 1:2]
            centre = centre + shift
        else:
            raise NotImplementedError(head.add_gt_groups_noise)
        x = centre[:, 0].clip(min=pcr[0] + 1e-6, max=pcr[3] - 1e-5)
        y = centre[:, 1].clip(min=pcr[1] + 1e-6, max=pcr[4] - 1e-5)
        gx = ((x - pcr[0]) / (pcr[3] - pcr[0]) * cols).clip(max=cols - 1, min=0).to(torch.int64)
        gy = ((y - pcr[1]) / (pcr[4] - pcr[1]) * rows).clip(max=rows - 1, min=0).to(torch.int64)
        lab = torch.where(shift.norm(dim=1) < head.add_gt_pos_thresh, lab, torch.full_like(lab, K))
        if kind.startswith('box'):
            lab = torch.where(noise.norm(dim=1) < head.add_gt_pos_boxnoise_thresh, lab, torch.full_like(lab, K))
        cells.append(gy * cols + gx)
        labels_out.append(lab)
    cell = torch.stack(cells)                                                      # (B, M*G) flat BEV cell of each gt query
    gt_labels = torch.stack(labels_out)
    gt_pos = bev_pos.expand(B, -1, -1).gather(1, cell[..., None].expand(-1, -1, bev_pos.shape[-1]))
    gt_score = heatmap.gather(-1, cell[:, None, :].expand(-1, K, -1))
    gt_feat = lidar_feat_flatten.gather(-1, cell[:, None, :].expand(-1, lidar_feat_flatten.shape[1], -1))
    if len(spec) > 2 and spec[2] == 'heatmap':
        one_hot = F.one_hot(gt_score.argmax(1), num_classes=K + 1).permute(0, 2, 1)
    elif len(spec) > 2 and spec[2] == 'heatmapcls':
        one_hot = gt_score
    else:
        one_hot = F.one_hot(gt_labels, num_classes=K + 1).permute(0, 2, 1)
    gt_feat = gt_feat + head.class_encoding(one_hot[:, :K].float())
    vf = valid.float()
    query_pos = torch.cat([query_pos, gt_pos * vf[..., None]], 1)
    query_feat = torch.cat([query_feat, gt_feat * vf[:, None, :]], 2)
    query_heatmap_score = torch.cat([query_heatmap_score, gt_score * vf[:, None, :]], 2)
    return query_feat, query_pos, query_heatmap_score, valid, gt_labels


def roi_grid(head, query_box, stage, dataset):
    """FD:890-910: the g x g sample points of every (expanded) box, normalised to [-1, 1] over the hard-coded range.
    query_box (B, >=8, Nq) raw predictions -> (B, Nq, g*g, 2) (x, y)."""
    from .focal_decoder import _ROI_RANGE
    osf, vx, vy, px, py = head.bbox_coder.coder_params
    g, ratio = head.roi_feats, head.roi_expand_ratio[stage]
    cx = query_box[:, 0] * (osf * vx) + px                                         # decode_box, BC:54-69
    cy = query_box[:, 1] * (osf * vy) + py
    sx, sy = (query_box[:, 3] * ratio).exp(), (query_box[:, 4] * ratio).exp()
    yaw = torch.atan2(query_box[:, 6], query_box[:, 7])
    i = torch.arange(g, device=query_box.device, dtype=torch.float32)
    u = ((i + 0.5) / g - 0.5)
    ux, uy = u.repeat_interleave(g), u.repeat(g)                                   # FD:1655-1664: first index varies slowest
    x, y = ux * sx[..., None], uy * sy[..., None]                                  # (B, Nq, g*g)
    s, c = torch.sin(yaw)[..., None], torch.cos(yaw)[..., None]
    gx, gy = x * c + y * s + cx[..., None], -x * s + y * c + cy[..., None]         # rotation_3d_in_axis, Appendix A.5
    x0, y0, x1, y1 = _ROI_RANGE[dataset]
    grid = torch.stack([(gx - x0) / (x1 - x0), (gy - y0) / (y1 - y0)], -1) * 2. - 1.
    return grid.clip(min=-2., max=2.)


def roi_features(head, flat_cl, levels, level_hw, query_box, stage, dataset):
    """FD:890-922 -> (B*Nq, C): RoI matrix (every level sampled on the g x g grid of the previous stage's box) through roi_mlp.
    Default: ``RoIGridSampleFunction`` on the channels-last pyramid (HIP forward + backward, columns [level][point][channel],
    the first Linear's columns permuted to match under autograd).  ``head.train_roi_sampler = 'grid_sample'`` keeps the
    reference's own op sequence on the framework's grid_sample (the parity tests run both)."""
    from .autograd import RoIGridSampleFunction
    from .focal_decoder import _ROI_RANGE
    B, Nn = query_box.shape[0], query_box.shape[2]
    C, G = flat_cl.shape[-1], head.roi_feats ** 2
    if getattr(head, 'train_roi_sampler', 'hip') == 'grid_sample':
        grid = roi_grid(head, query_box, stage, dataset)
        roi = torch.cat([F.grid_sample(f, grid, mode='bilinear', align_corners=False) for f in levels], 1)
        return head.roi_mlp(roi.permute(0, 2, 1, 3).reshape(B * Nn, -1))      # columns [level][channel][point], FD:919
    roi = RoIGridSampleFunction.apply(flat_cl, query_box, level_hw, head.roi_feats, head.roi_expand_ratio[stage],
                                      head.bbox_coder.coder_params, _ROI_RANGE[dataset], 1)
    lin = head.roi_mlp[0]
    w = lin.weight.view(lin.weight.shape[0], len(levels), C, G).permute(0, 1, 3, 2).reshape(lin.weight.shape[0], -1)
    y = F.linear(roi, w, lin.bias)
    for m in list(head.roi_mlp)[1:]:
        y = m(y)
    return y


def forward_train(head, pts_inputs, gt_bboxes_3d=None, gt_labels_3d=None):
    """FD:522-992 with ``self.training`` -> the prediction dict (see module docstring)."""
    if getattr(head, 'heatmap_box', False):
        raise NotImplementedError('FocalDecoder: the heatmap_box branch is built for inference only (.eval()); its training side '
                                  '(noised ground-truth boxes FD:489-516, CenterPoint-style targets FD:1415-1653, separate losses '
                                  'FD:1253-1309) is not - no shipped config enables the branch')
    head.num_proposals = head.num_proposals_ori
    lidar_feat = pts_inputs[0]
    second, extra = pts_inputs[1], None
    if head.extra_feat:
        extra, second = second[-1], list(second[:-1])
    B, C, H, W = lidar_feat.shape
    K, k = head.num_classes, head.num_proposals_ori
    dataset = head.test_cfg['dataset']
    bits, ks = ops.small_class_bits(dataset, K), head.nms_kernel_size
    dev = lidar_feat.device
    n_st = int(head.multistage_heatmap or 0)
    Nq = k * max(n_st, 1)
    qpos = torch.empty(B, Nq, 2, device=dev)
    qscore = torch.empty(B, K, Nq, device=dev)
    qlabel = torch.empty(B, Nq, dtype=torch.int64, device=dev)
    scratch = torch.empty(B, Nq, C, device=dev)          # the kernel's own feature gather: unused, recomputed under autograd
    cls_w = head.class_encoding.weight.detach().view(C, K).contiguous()
    cls_b = head.class_encoding.bias.detach().contiguous()

    def query_features(feat, idx):
        """FD:692-699 under autograd: BEV feature at the selected cell + class embedding -> (B, C, k)."""
        cell, cls = idx % (H * W), idx // (H * W)
        q = feat.reshape(B, C, H * W).gather(-1, cell[:, None, :].expand(-1, C, -1))
        return q + head.class_encoding(F.one_hot(cls, num_classes=K).permute(0, 2, 1).float())

    masks_out, qfeats = [], []
    if not n_st:                                                                   # single-stage branch, FD:538-586
        dense = head.heatmap_head(lidar_feat)
        if head.input_img or head.iterbev_wo_img:
            new_feat = (second[-1] if isinstance(second, (list, tuple)) else second).reshape(lidar_feat.shape)
            dense_img = head.heatmap_head_img(new_feat)
            heat, hist, _ = ops.heatmap_nms(dense.detach().contiguous(), None, dense_img.detach().contiguous(), ks, bits,
                                            want_mask_next=False)
            heatmap_train = [dense, dense_img]
        else:
            new_feat = lidar_feat
            heat, hist, _ = ops.heatmap_nms(dense.detach().contiguous(), None, None, ks, bits, want_mask_next=False)
            heatmap_train = dense
        idx = ops.topk(heat, hist, k)
        ops.query_gather(new_feat.detach().contiguous(), heat, idx, cls_w, cls_b, scratch, qpos, qscore, qlabel, None, 0, 0,
                         ks, bits)
        qfeats.append(query_features(new_feat, idx))
        pyramid_src = flat_src = new_feat
    else:                                                                          # Hard Instance Probing, FD:587-791
        feats = list(second)
        if head.reuse_first_heatmap:
            feats.insert(0, lidar_feat)
        dense0 = head.heatmap_head(lidar_feat)
        logits = [dense0 if (i == 0 and head.reuse_first_heatmap) else head.heatmap_head_img[i](feats[i]) for i in range(n_st)]
        mask_mode = {'pos': 2, 'poscls': 1}.get(head.mask_heatmap_mode, 0)
        ones = torch.ones(B, K, H, W, device=dev)
        heatmap_train, mask = [], None
        for i in range(n_st):
            if i == 0:
                heatmap_train.append(dense0)
                masks_out.append(ones)
                if not head.reuse_first_heatmap:                                   # FD:663-668: two entries for stage 0
                    heatmap_train.append(logits[0])
                    masks_out.append(ones)
            else:
                heatmap_train.append(logits[i])
                masks_out.append(mask)
            last = i == n_st - 1
            heat, hist, nxt = ops.heatmap_nms(logits[i].detach().contiguous(), mask, None, ks, bits, want_mask_next=not last)
            idx = ops.topk(heat, hist, k)
            ops.query_gather(feats[i].detach().contiguous(), heat, idx, cls_w, cls_b, scratch, qpos, qscore, qlabel, nxt, i * k,
                             mask_mode if not last else 0, ks, bits)
            qfeats.append(query_features(feats[i], idx))
            mask = nxt
        head.num_proposals = Nq
        pyramid_src = extra if head.extra_feat else feats[-1]
        flat_src = feats[-1]
    head.query_labels = qlabel
    query_feat = torch.cat(qfeats, 2)                                              # (B, C, Nq)
    query_labels = qlabel

    groups = head.add_gt_groups if gt_labels_3d is not None else 0
    if gt_labels_3d is not None:                                                   # FD:793-795
        head.num_gts = [int(t.shape[0]) for t in gt_labels_3d]
        head.max_num_gts = max(head.num_gts)
    valid = gt_query_labels = None
    if groups > 0:                                                                 # FD:798-808
        bev_pos = _static(head, ('bev_pos', dev), lambda: head.bev_pos.to(dev))
        query_feat, qpos, qscore, valid, gt_query_labels = head.generate_gt_groups(
            query_feat, qpos, qscore, lidar_feat, flat_src.reshape(B, C, -1), bev_pos, heat.view(B, K, -1),
            gt_bboxes_3d, gt_labels_3d)
        query_labels = torch.cat([query_labels, gt_query_labels], 1)
    n_gt = head.max_num_gts * groups if groups > 0 else 0
    Nn = Nq + n_gt

    levels = [pyramid_src, ] if head.multiscale else [flat_src]                    # BEV pyramid, FD:810-823
    if head.multiscale:
        levels.append(head.dconv(levels[-1]))
        levels.append(head.dconv2(levels[-1]))
    level_hw = [tuple(f.shape[2:]) for f in levels]
    Hs, Ws = level_hw[0]
    wh = _static(head, ('wh', dev, Ws, Hs), lambda: torch.tensor([float(Ws), float(Hs)], device=dev))
    flat = torch.cat([f.flatten(2, 3) for f in levels], -1).transpose(1, 2).contiguous()    # (B, Nv, C) channels-last
    attn_mask = None
    if groups > 0:                                                                 # FD:849-856
        attn_mask = torch.ones(B, Nn, Nn, dtype=torch.bool, device=dev)
        attn_mask[:, :, :Nq] = False                                               # every query sees the heatmap queries
        attn_mask[:, Nq:, Nq:] = ~(valid[:, None] & valid[:, :, None])             # gt queries see the valid gt queries
        # (one (Nn, Nn) mask per frame; FD:856 repeats it over the heads - MultiheadAttention.forward_train_bf takes the
        #  per-frame form as it is and expands only for the framework's own attention routes)
    if head.bevpos:                                      # the sine embedding of the pyramid's cell centres: configuration only
        def make_bev_sine():
            grids = [head.create_2D_grid(h, w) * float(2 ** l) for l, (h, w) in enumerate(level_hw)]
            return gen_sineembed_for_position(torch.cat(grids, 1)[0].to(dev).contiguous(), float(Ws), float(Hs))
        bev_sine = _static(head, ('bev_sine', dev, tuple(level_hw)), make_bev_sine)

    ret, query_box = [], None
    x = query_feat.transpose(1, 2)                                                 # (B, Nn, C) batch-first from here on
    for s in range(head.num_decoder_layers):
        ref = qpos / wh                                                            # FD:869
        qpe = head.pos_embed_learned[s](gen_sineembed_for_position(qpos.contiguous(), float(Ws), float(Hs)))
        value = flat + head.pos_embed_learned[s](bev_sine)[None] if head.bevpos else flat       # FD:883-888
        if head.roi_feats and query_box is not None:                               # FD:890-922
            x = x + roi_features(head, flat, levels, level_hw, query_box, s, dataset).view(B, Nn, C)
        x = head.decoder[s].forward_bf(x, value, qpe, ref, level_hw, attn_mask)    # FD:927-933
        res = head.prediction_heads[s](x.transpose(1, 2))                          # FD:939
        if head.classaware_reg:                                                    # FD:940-943
            for key in ('center', 'height', 'dim', 'rot'):
                r_ = res[key].reshape(B, K, -1, Nn)
                res[key] = r_.gather(1, query_labels[:, None, None, :].expand(-1, -1, r_.shape[2], -1).clip(0, K - 1))[:, 0]
        res['center'] = res['center'] + (ref * wh).transpose(1, 2)                 # FD:936,945
        qpos = res['center'].detach().transpose(1, 2).contiguous()                 # FD:947
        if head.roi_based_reg and query_box is not None:                           # FD:949-951
            res['dim'] = torch.cat([res['dim'][:, :2] + query_box[:, 3:5], res['dim'][:, 2:]], 1)
            res['rot'] = res['rot'] + query_box[:, 6:8]
        parts = [res['center'], res['height'], res['dim'], res['rot']] + ([res['vel']] if 'vel' in res else [])
        query_box = torch.cat(parts, 1).detach()                                   # FD:956
        ret.append(res)

    out = {}
    for key in ret[0]:                                                             # FD:970-987
        if n_gt:
            out[key] = torch.cat([r[key][:, :, :-n_gt] for r in ret], -1)
            out[key + '_gtgroups'] = torch.cat([r[key][:, :, -n_gt:] for r in ret], -1)
        else:
            out[key] = torch.cat([r[key] for r in ret], -1)
    if n_gt:
        out['batch_valid_gt_mask'], out['batch_gt_query_labels'] = valid, gt_query_labels
    out['query_heatmap_score'] = qscore
    out['dense_heatmap'] = heatmap_train
    if n_st:
        out['multistage_masks'] = masks_out
    return out
