"""CPU oracle of the head's TRAINING targets and losses - test infrastructure, never imported by
the product (only tests/ and oracle/gen_golden.py use it).

Two kinds of code live here:

 1. restatements of the un-vendored third-party pieces the reference builds from its config (source absent from
    /root/reference; pinned versions mmdet 2.14.0 / mmdet3d 0.17.1, doc/install.md:9-14) - "parity unpinned by execution":
      FocalLossCost, FocalLoss / L1Loss / GaussianFocalLoss (+ weight_reduce_loss), AssignResult, PseudoSampler,
      BboxOverlaps3D (3-D IoU = rotated BEV overlap x height overlap), gaussian_radius / draw_heatmap_gaussian,
      clip_sigmoid, multi_apply.
    oracle/ref_shims.py serves exactly these objects to the reference's own code when the golden fixtures are generated.
 2. a restatement of the reference's own algorithm - HungarianAssigner3D.assign (core/bbox/assigners/
    hungarian_assigner.py:97-162) with its match costs (:15-47), FocalDecoder.get_targets_single / get_targets
    (dense_heads/focal_decoder.py:1022-1164 / 994-1020) and FocalDecoder.loss (:1166-1311; without the heatmap_box branch,
    which no shipped config enables) - pinned by tests/golden/train_targets.npz, which the reference's code produced.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import ff3d_oracle as O


# --------------------------------------------------------------------------- 1. un-vendored third party (restated)
def boxes_iou3d(a, b):
    """mmdet3d `BboxOverlaps3D(coordinate='lidar')(a, b)` = `LiDARInstance3DBoxes.overlaps(mode='iou')`: boxes
    (x, y, z_bottom, dx, dy, dz, yaw, ...) -> (N, M) IoU of the 3-D boxes."""
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    if len(a) == 0 or len(b) == 0:
        return a.new_zeros(len(a), len(b))
    bev_a, bev_b = O.xywhr2xyxyr(a[:, [0, 1, 3, 4, 6]]), O.xywhr2xyxyr(b[:, [0, 1, 3, 4, 6]])
    overlap_bev = torch.from_numpy(O.boxes_overlap_bev(bev_a.numpy(), bev_b.numpy()))
    top = torch.min((a[:, 2] + a[:, 5])[:, None], (b[:, 2] + b[:, 5])[None, :])
    bottom = torch.max(a[:, 2][:, None], b[:, 2][None, :])
    overlap_3d = overlap_bev * (top - bottom).clamp(min=0)
    vol_a, vol_b = (a[:, 3] * a[:, 4] * a[:, 5])[:, None], (b[:, 3] * b[:, 4] * b[:, 5])[None, :]
    return overlap_3d / (vol_a + vol_b - overlap_3d).clamp(min=1e-8)


class BboxOverlaps3D:
    def __init__(self, coordinate='lidar', **kw):
        assert coordinate == 'lidar'

    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        assert mode == 'iou' and not is_aligned
        return boxes_iou3d(bboxes1, bboxes2).to(bboxes1.device)


class FocalLossCost:
    """mmdet 2.14 `FocalLossCost(weight, alpha, gamma, eps)`: cls_pred (num_query, num_class) logits, gt_labels (num_gt)."""

    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        cls_pred = cls_pred.sigmoid()
        neg_cost = -(1 - cls_pred + self.eps).log() * (1 - self.alpha) * cls_pred.pow(self.gamma)
        pos_cost = -(cls_pred + self.eps).log() * self.alpha * (1 - cls_pred).pow(self.gamma)
        return (pos_cost[:, gt_labels] - neg_cost[:, gt_labels]) * self.weight


class AssignResult:
    """mmdet `AssignResult(num_gts, gt_inds, max_overlaps, labels)`: gt_inds 0 = background, k > 0 = gt k-1."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class SamplingResult:
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_bboxes.numel() == 0:
            self.pos_gt_bboxes = torch.empty_like(gt_bboxes).view(-1, gt_bboxes.shape[-1] if gt_bboxes.dim() == 2 else 4)
        else:
            self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds.long(), :]


class PseudoSampler:
    """mmdet `PseudoSampler.sample`: every assigned box is a sample."""

    def __init__(self, **kw):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kw):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result)


def multi_apply(func, *args, **kwargs):
    """mmdet `multi_apply`."""
    res = map(lambda *a: func(*a, **kwargs), *args)
    return tuple(map(list, zip(*res)))


def clip_sigmoid(x, eps=1e-4):
    """mmdet3d `clip_sigmoid`."""
    return torch.clamp(x.sigmoid(), min=eps, max=1 - eps)


def gaussian_2d(shape, sigma=1):
    m, n = [(ss - 1.) / 2. for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_heatmap_gaussian(heatmap, center, radius, k=1):
    """mmdet3d `draw_heatmap_gaussian` (core/utils/gaussian.py): in-place max with a (2r+1)^2 Gaussian, sigma = (2r+1)/6."""
    diameter = 2 * radius + 1
    gaussian = gaussian_2d((diameter, diameter), sigma=diameter / 6)
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[0:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    masked_heatmap = heatmap[y - top:y + bottom, x - left:x + right]
    masked_gaussian = torch.from_numpy(gaussian[radius - top:radius + bottom, radius - left:radius + right]).to(
        heatmap.device, torch.float32)
    if min(masked_gaussian.shape) > 0 and min(masked_heatmap.shape) > 0:
        torch.max(masked_heatmap, masked_gaussian * k, out=masked_heatmap)
    return heatmap


def gaussian_radius(det_size, min_overlap=0.5):
    """mmdet3d `gaussian_radius` (the CenterNet formula, tensor arithmetic)."""
    height, width = det_size
    a1 = 1
    b1 = (height + width)
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    sq1 = torch.sqrt(b1 ** 2 - 4 * a1 * c1)
    r1 = (b1 + sq1) / 2
    a2 = 4
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    sq2 = torch.sqrt(b2 ** 2 - 4 * a2 * c2)
    r2 = (b2 + sq2) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    sq3 = torch.sqrt(b3 ** 2 - 4 * a3 * c3)
    r3 = (b3 + sq3) / 2
    return min(r1, r2, r3)


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    """mmdet `weight_reduce_loss`."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == 'mean' else loss.sum() if reduction == 'sum' else loss
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


class FocalLoss:
    """mmdet 2.14 `FocalLoss(use_sigmoid=True)` with integer targets (num_classes = background)."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0, **kw):
        assert use_sigmoid
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, loss_weight

    def __call__(self, pred, target, weight=None, avg_factor=None):
        num_classes = pred.size(1)
        t = F.one_hot(target, num_classes=num_classes + 1)[:, :num_classes].type_as(pred)
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        focal_weight = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction='none') * focal_weight
        if weight is not None and weight.shape != loss.shape:
            weight = weight.view(-1, 1) if weight.size(0) == loss.size(0) else weight.view(loss.size(0), -1)
        return self.loss_weight * weight_reduce_loss(loss, None if weight is None else weight.float(), self.reduction,
                                                     avg_factor)


class L1Loss:
    def __init__(self, reduction='mean', loss_weight=1.0, **kw):
        self.reduction, self.loss_weight = reduction, loss_weight

    def __call__(self, pred, target, weight=None, avg_factor=None):
        return self.loss_weight * weight_reduce_loss((pred - target).abs(), weight, self.reduction, avg_factor)


class GaussianFocalLoss:
    def __init__(self, alpha=2.0, gamma=4.0, reduction='mean', loss_weight=1.0, **kw):
        self.alpha, self.gamma, self.reduction, self.loss_weight = alpha, gamma, reduction, loss_weight

    def __call__(self, pred, target, weight=None, avg_factor=None):
        eps = 1e-12
        pos_weights = target.eq(1)
        neg_weights = (1 - target).pow(self.gamma)
        pos_loss = -(pred + eps).log() * (1 - pred).pow(self.alpha) * pos_weights
        neg_loss = -(1 - pred + eps).log() * pred.pow(self.alpha) * neg_weights
        return self.loss_weight * weight_reduce_loss(pos_loss + neg_loss, weight, self.reduction, avg_factor)


LOSSES = {'FocalLoss': FocalLoss, 'L1Loss': L1Loss, 'GaussianFocalLoss': GaussianFocalLoss}


def build_loss(cfg):
    cfg = dict(cfg)
    return LOSSES[cfg.pop('type')](**cfg)


# --------------------------------------------------------------------------- 2. the reference's algorithm (restated)
def bbox_bev_l1_cost(bboxes, gt_bboxes, point_cloud_range, weight):
    """hungarian_assigner.py:25-37 BBoxBEVL1Cost."""
    pc_start = bboxes.new_tensor(point_cloud_range[0:2])
    pc_range = bboxes.new_tensor(point_cloud_range[3:5]) - pc_start
    a, b = (bboxes[:, :2] - pc_start) / pc_range, (gt_bboxes[:, :2] - pc_start) / pc_range
    return torch.cdist(a, b, p=1) * weight


def hungarian_assign(bboxes, gt_bboxes, gt_labels, cls_pred, train_cfg):
    """hungarian_assigner.py:111-162 -> (gt_inds (n,), max_overlaps (n,), labels (n,))."""
    from scipy.optimize import linear_sum_assignment
    a = train_cfg['assigner']
    n, m = bboxes.size(0), gt_bboxes.size(0)
    gt_inds = bboxes.new_full((n,), -1, dtype=torch.long)
    labels = bboxes.new_full((n,), -1, dtype=torch.long)
    if m == 0 or n == 0:
        if m == 0:
            gt_inds[:] = 0
        return gt_inds, None, labels
    cc = dict(a['cls_cost'])
    assert cc.pop('type') == 'FocalLossCost'
    cls_cost = FocalLossCost(**cc)(cls_pred[0].T, gt_labels)
    assert a['reg_cost']['type'] == 'BBoxBEVL1Cost'
    reg_cost = bbox_bev_l1_cost(bboxes, gt_bboxes, train_cfg['point_cloud_range'], a['reg_cost']['weight'])
    iou = boxes_iou3d(bboxes, gt_bboxes)
    cost = cls_cost + reg_cost - iou * a['iou_cost']['weight']
    rows, cols = linear_sum_assignment(cost.detach().cpu())
    rows, cols = torch.from_numpy(rows), torch.from_numpy(cols)
    gt_inds[:] = 0
    gt_inds[rows] = cols + 1
    labels[rows] = gt_labels[cols]
    max_overlaps = torch.zeros_like(iou.max(1).values)
    max_overlaps[rows] = iou[rows, cols]
    return gt_inds, max_overlaps, labels


def heuristic_assign(bboxes, gt_bboxes, gt_labels, query_labels=None, dist_thre=100):
    """HeuristicAssigner3D.assign (core/bbox/assigners/hungarian_assigner.py:58-91), the reference's loop as it stands:
    every ground-truth box visits its nearest proposal in ascending box order; a proposal keeps the nearest box that claimed
    it (strict <).  -> (gt_inds long 0 = background / k = box k-1, max_overlaps, labels float, -1 = none)."""
    num_gts, num_bboxes = len(gt_bboxes), len(bboxes)
    bev_dist = torch.norm(bboxes[:, 0:2][None, :, :] - gt_bboxes[:, 0:2][:, None, :], dim=-1)      # (num_gts, num_bboxes)
    if query_labels is not None:
        bev_dist = bev_dist + (query_labels[None] != gt_labels[:, None]) * dist_thre               # :65-66
    _, nearest = bev_dist.min(1)                                                                    # :69
    inds = torch.zeros(num_bboxes)
    vals = torch.full((num_bboxes,), 10000.0)
    labels = torch.full((num_bboxes,), -1.0)
    for g in range(num_gts):                                                                        # :73-80
        p = int(nearest[g])
        if bev_dist[g, p] <= dist_thre and bev_dist[g, p] < vals[p]:
            vals[p], inds[p], labels[p] = bev_dist[g, p], g + 1, float(gt_labels[g])
    overlaps = torch.zeros(num_bboxes)
    m = torch.where(inds > 0)[0]
    if len(m):
        overlaps[m] = boxes_iou3d(gt_bboxes[inds[m].long() - 1], bboxes[m]).diag()                 # :83-85
    return inds.long(), overlaps, labels


def encode_boxes(dst, pc_range, voxel_size, out_size_factor, code_size):
    """BC:24-37."""
    t = torch.zeros(dst.shape[0], code_size)
    t[:, 0] = (dst[:, 0] - pc_range[0]) / (out_size_factor * voxel_size[0])
    t[:, 1] = (dst[:, 1] - pc_range[1]) / (out_size_factor * voxel_size[1])
    t[:, 3:6] = (dst[:, 3:6] + 1e-6).log()
    t[:, 2] = dst[:, 2] + dst[:, 5] * 0.5
    t[:, 6], t[:, 7] = torch.sin(dst[:, 6]), torch.cos(dst[:, 6])
    if code_size == 10:
        t[:, 8:10] = dst[:, 7:]
    return t


def get_targets_single(gt_boxes, gt_labels, preds, cfg, train_cfg, num_proposals, num_decoder_layers, num_classes,
                       code_size, gt_center_limit=None):
    """FD:1022-1164 for one sample.  gt_boxes (m, 7|9) LiDAR boxes (bottom centre), preds: dict of (1, c, n_all) tensors."""
    n_all = preds['center'].shape[-1]
    score = preds['heatmap'].detach().clone()
    _, (boxes_all, _, _, _) = O.bbox_decode(score, preds['rot'].detach().clone(), preds['dim'].detach().clone(),
                                            preds['center'].detach().clone(), preds['height'].detach().clone(),
                                            preds['vel'].detach().clone() if 'vel' in preds else None, cfg)
    boxes = boxes_all[0]                                             # decode(filter=False): every proposal (BC:143-150)
    gi, mo, lb = [], [], []
    for l in range(num_decoder_layers):
        sl = slice(num_proposals * l, num_proposals * (l + 1))
        g, m, la = hungarian_assign(boxes[sl], gt_boxes, gt_labels, score[..., sl], train_cfg)
        if gt_center_limit is not None:
            pos = g > 0
            bad = (gt_boxes[g[pos] - 1][:, :2] - boxes[sl][pos][:, :2]).norm(dim=1) > gt_center_limit
            g[torch.nonzero(pos)[:, 0][bad]] = 0
        gi.append(g), mo.append(m), lb.append(la)
    gt_inds, max_overlaps = torch.cat(gi), torch.cat(mo)
    pos_inds = torch.nonzero(gt_inds > 0).squeeze(-1).unique()
    neg_inds = torch.nonzero(gt_inds == 0).squeeze(-1).unique()
    assert len(pos_inds) + len(neg_inds) == n_all
    bbox_targets, bbox_weights = torch.zeros(n_all, code_size), torch.zeros(n_all, code_size)
    ious = max_overlaps.clamp(0.0, 1.0)
    labels = boxes.new_zeros(n_all, dtype=torch.long) + num_classes
    label_weights = boxes.new_zeros(n_all, dtype=torch.long)
    if len(pos_inds) > 0:
        pos_gt = gt_boxes[(gt_inds[pos_inds] - 1).long()]
        bbox_targets[pos_inds] = encode_boxes(pos_gt, cfg.pc_range, cfg.voxel_size, cfg.out_size_factor, code_size)
        bbox_weights[pos_inds] = 1.0
        labels[pos_inds] = gt_labels[gt_inds[pos_inds] - 1]
        label_weights[pos_inds] = 1 if train_cfg['pos_weight'] <= 0 else train_cfg['pos_weight']
    if len(neg_inds) > 0:
        label_weights[neg_inds] = 1
    # dense heatmap target (FD:1133-1158); gravity centre = bottom centre + dz / 2
    g3 = torch.cat([gt_boxes[:, :2], (gt_boxes[:, 2] + gt_boxes[:, 5] * 0.5)[:, None], gt_boxes[:, 3:]], 1)
    grid_size, pc_range = torch.tensor(train_cfg['grid_size']), torch.tensor(train_cfg['point_cloud_range'])
    voxel_size, osf = torch.tensor(train_cfg['voxel_size']), train_cfg['out_size_factor']
    fmap = grid_size[:2] // osf
    heatmap = g3.new_zeros(num_classes, int(fmap[1]), int(fmap[0]))
    for i in range(len(g3)):
        width, length = g3[i][3] / voxel_size[0] / osf, g3[i][4] / voxel_size[1] / osf
        if width > 0 and length > 0:
            radius = gaussian_radius((length, width), min_overlap=train_cfg['gaussian_overlap'])
            radius = max(train_cfg['min_radius'], int(radius))
            cx = (g3[i][0] - pc_range[0]) / voxel_size[0] / osf
            cy = (g3[i][1] - pc_range[1]) / voxel_size[1] / osf
            center_int = torch.tensor([cx, cy], dtype=torch.float32).to(torch.int32)
            draw_heatmap_gaussian(heatmap[gt_labels[i]], center_int, radius)
    mean_iou = ious[pos_inds].sum() / max(len(pos_inds), 1)
    return (labels[None], label_weights[None], bbox_targets[None], bbox_weights[None], ious[None], int(pos_inds.shape[0]),
            float(mean_iou), heatmap[None])


def get_targets(gt_boxes_list, gt_labels_list, preds, cfg, train_cfg, **kw):
    """FD:994-1020."""
    res = [get_targets_single(g, l, {k: v[b:b + 1] for k, v in preds.items() if torch.is_tensor(v)}, cfg, train_cfg, **kw)
           for b, (g, l) in enumerate(zip(gt_boxes_list, gt_labels_list))]
    cat = lambda i: torch.cat([r[i] for r in res], 0)                                    # noqa: E731
    return cat(0), cat(1), cat(2), cat(3), cat(4), int(np.sum([r[5] for r in res])), float(np.mean([r[6] for r in res])), cat(7)


def head_loss(gt_boxes_list, gt_labels_list, preds, cfg, train_cfg, loss_cfgs, num_proposals, num_decoder_layers, num_classes,
              code_size, loss_weight_heatmap=1.0, gt_center_limit=None):
    """FD:1166-1311 without the gt-group and heatmap_box terms.  preds: the head's output dict (one stage list element)."""
    labels, label_weights, bbox_targets, bbox_weights, ious, num_pos, matched_ious, heatmap = get_targets(
        gt_boxes_list, gt_labels_list, preds, cfg, train_cfg, num_proposals=num_proposals,
        num_decoder_layers=num_decoder_layers, num_classes=num_classes, code_size=code_size, gt_center_limit=gt_center_limit)
    loss_cls, loss_bbox, loss_hm = (build_loss(loss_cfgs[k]) for k in ('loss_cls', 'loss_bbox', 'loss_heatmap'))
    out = {}
    dense = preds['dense_heatmap']
    if isinstance(dense, (tuple, list)):
        masks = torch.cat(preds['multistage_masks'], 0) if 'multistage_masks' in preds else None
        hm = heatmap.repeat(len(dense), 1, 1, 1)
        if masks is not None:
            hm = hm * masks
        out['loss_heatmap'] = loss_hm(clip_sigmoid(torch.cat(list(dense), 0)), hm, weight=masks,
                                      avg_factor=max(hm.eq(1).float().sum().item(), 1)) * loss_weight_heatmap
    else:
        out['loss_heatmap'] = loss_hm(clip_sigmoid(dense), heatmap,
                                      avg_factor=max(heatmap.eq(1).float().sum().item(), 1)) * loss_weight_heatmap
    code_weights = train_cfg.get('code_weights', None)
    for l in range(num_decoder_layers):
        sl = slice(l * num_proposals, (l + 1) * num_proposals)
        cls_score = preds['heatmap'][..., sl].permute(0, 2, 1).reshape(-1, num_classes)
        out[f'layer_{l}_loss_cls'] = loss_cls(cls_score, labels[..., sl].reshape(-1), label_weights[..., sl].reshape(-1),
                                              avg_factor=max(num_pos, 1))
        parts = [preds[k][..., sl] for k in ('center', 'height', 'dim', 'rot')] + ([preds['vel'][..., sl]] if 'vel' in preds else [])
        p = torch.cat(parts, 1).permute(0, 2, 1)
        w = bbox_weights[:, sl, :] * bbox_weights.new_tensor(code_weights)
        out[f'layer_{l}_loss_bbox'] = loss_bbox(p, bbox_targets[:, sl, :], w, avg_factor=max(num_pos, 1))
    out['matched_ious'] = torch.tensor(matched_ious)
    return out
