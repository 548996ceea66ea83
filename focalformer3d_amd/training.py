"""Training targets and losses of the FocalDecoder head on MI355X - first slice of the training path.

Mirrors, with the reference's registry names and call signatures:

  HungarianAssigner3D, HeuristicAssigner3D, BBox3DL1Cost, BBoxBEVL1Cost, IoU3DCost   core/bbox/assigners/hungarian_assigner.py:15-162
  FocalLossCost, BboxOverlaps3D, AssignResult, PseudoSampler,    un-vendored mmdet 2.14 / mmdet3d 0.17.1 pieces the
  FocalLoss, L1Loss, GaussianFocalLoss, clip_sigmoid             reference builds from its config (restated)
  head_get_targets_single / head_get_targets / head_loss         FocalDecoder.get_targets_single / get_targets / loss,
                                                                 dense_heads/focal_decoder.py:1022-1164, 994-1020, 1166-1311

Device mapping: predictions and targets stay on the MI355X.  The matching cost is built on the device - classification and
BEV-L1 terms as tensor ops, the 3-D IoU matrix by the HIP kernel ``ff3d_boxes_iou3d`` (rotated BEV overlap x height overlap,
mmdet3d ``BboxOverlaps3D``) - and, exactly as in the reference (hungarian_assigner.py:144-151), only the small (proposals x
gts) cost matrix goes to the host for scipy's ``linear_sum_assignment``.  The dense heatmap target is one launch of
``ff3d_gaussian_heatmap_targets`` per sample instead of the reference's per-box host loop (FD:1141-1158).  Losses are
differentiable tensor expressions.  Not mirrored: the training-mode forward (ground-truth query groups FD:377-520, attention
masks FD:849-858, dropout) - ``add_gt_groups`` terms of the loss are computed only when the predictions carry them.
"""
import copy

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .registry import (BBOX_ASSIGNERS, IOU_CALCULATORS, LOSSES, MATCH_COST, build_iou_calculator, build_match_cost, register)

try:
    from scipy.optimize import linear_sum_assignment
except ImportError:                                              # pragma: no cover
    linear_sum_assignment = None


# ------------------------------------------------------------------------------------------------ match costs
@register(MATCH_COST)
class BBox3DL1Cost:
    """hungarian_assigner.py:15-22."""

    def __init__(self, weight):
        self.weight = weight

    def __call__(self, bboxes, gt_bboxes, train_cfg):
        return torch.cdist(bboxes, gt_bboxes, p=1) * self.weight


@register(MATCH_COST)
class BBoxBEVL1Cost:
    """hungarian_assigner.py:25-37: L1 distance of the box centres normalised by the point-cloud range."""

    def __init__(self, weight):
        self.weight = weight

    def __call__(self, bboxes, gt_bboxes, train_cfg):
        pc_start = bboxes.new_tensor(train_cfg['point_cloud_range'][0:2])
        pc_range = bboxes.new_tensor(train_cfg['point_cloud_range'][3:5]) - pc_start
        a, b = (bboxes[:, :2] - pc_start) / pc_range, (gt_bboxes[:, :2] - pc_start) / pc_range
        return torch.cdist(a, b, p=1) * self.weight


@register(MATCH_COST)
class IoU3DCost:
    """hungarian_assigner.py:40-47."""

    def __init__(self, weight):
        self.weight = weight

    def __call__(self, iou):
        return -iou * self.weight


@register(MATCH_COST)
class FocalLossCost:
    """mmdet 2.14 ``FocalLossCost``: cls_pred (num_query, num_class) logits, gt_labels (num_gt) -> (num_query, num_gt)."""

    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        p = cls_pred.sigmoid()
        neg = -(1 - p + self.eps).log() * (1 - self.alpha) * p.pow(self.gamma)
        pos = -(p + self.eps).log() * self.alpha * (1 - p).pow(self.gamma)
        return (pos[:, gt_labels] - neg[:, gt_labels]) * self.weight


@register(IOU_CALCULATORS)
class BboxOverlaps3D:
    """mmdet3d ``BboxOverlaps3D(coordinate='lidar')``: 3-D IoU matrix on the HIP kernel (ops.boxes_iou3d)."""

    def __init__(self, coordinate='lidar'):
        if coordinate != 'lidar':
            raise NotImplementedError("only the 'lidar' box convention is used by FocalFormer3D")
        self.coordinate = coordinate

    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        if mode != 'iou' or is_aligned:
            raise NotImplementedError("BboxOverlaps3D: mode='iou', is_aligned=False (the assigner's use)")
        return ops.boxes_iou3d(bboxes1.float().contiguous(), bboxes2.float().contiguous())


class AssignResult:
    """mmdet ``AssignResult``: gt_inds 0 = background, k > 0 = ground truth k-1 (-1 = ignore)."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class SamplingResult:
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        self.pos_gt_bboxes = (gt_bboxes[self.pos_assigned_gt_inds.long(), :] if gt_bboxes.numel()
                              else gt_bboxes.new_zeros(0, gt_bboxes.shape[-1]))


class PseudoSampler:
    """mmdet ``PseudoSampler``: every assigned proposal is a sample."""

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        return SamplingResult(pos, neg, bboxes, gt_bboxes, assign_result)


@register(BBOX_ASSIGNERS)
class HungarianAssigner3D:
    """hungarian_assigner.py:97-162: one-to-one matching of the proposals of one decoder stage to the ground truth."""

    def __init__(self, cls_cost=dict(type='ClassificationCost', weight=1.), reg_cost=dict(type='BBoxBEVL1Cost', weight=1.0),
                 iou_cost=dict(type='IoU3DCost', weight=1.0), iou_calculator=dict(type='BboxOverlaps3D')):
        self.cls_cost = build_match_cost(cls_cost)
        self.reg_cost = build_match_cost(reg_cost)
        self.iou_cost = build_match_cost(iou_cost)
        self.iou_calculator = build_iou_calculator(iou_calculator)

    def cost_matrix(self, bboxes, gt_bboxes, gt_labels, scores, train_cfg):
        """(matching cost, 3-D IoU), both (proposals, gts), on the device: classification + BEV-L1 + IoU terms
        (hungarian_assigner.py:118-131).  scores (proposals, K) raw logits."""
        iou = self.iou_calculator(bboxes, gt_bboxes)
        return self.cls_cost(scores, gt_labels) + self.reg_cost(bboxes, gt_bboxes, train_cfg) + self.iou_cost(iou), iou

    @staticmethod
    def match(cost_host):
        """The host step (hungarian_assigner.py:144-151): scipy's optimal one-to-one matching of a (proposals, gts) cost matrix
        -> int64 vector, entry p = 1 + the ground truth matched to proposal p, 0 = background."""
        if linear_sum_assignment is None:
            raise ImportError('Please run "pip install scipy" to install scipy first.')
        gt_inds = np.zeros(cost_host.shape[0], np.int64)
        rows, cols = linear_sum_assignment(cost_host)
        gt_inds[rows] = cols + 1
        return gt_inds

    def assign(self, bboxes, gt_bboxes, gt_labels, cls_pred, train_cfg):
        """mmdet's per-sample entry point (one decoder stage of one frame); the training loop itself goes through
        ``head_get_targets_batched``, which shares ``cost_matrix`` / ``match`` and visits the host once per batch."""
        P, G = bboxes.size(0), gt_bboxes.size(0)
        labels = bboxes.new_full((P,), -1, dtype=torch.long)
        if P == 0 or G == 0:
            return AssignResult(G, bboxes.new_full((P,), 0 if G == 0 else -1, dtype=torch.long), None, labels=labels)
        cost, iou = self.cost_matrix(bboxes, gt_bboxes, gt_labels, cls_pred[0].T, train_cfg)
        gt_inds = torch.from_numpy(self.match(cost.detach().cpu().numpy())).to(bboxes.device)
        hit = gt_inds > 0
        labels[hit] = gt_labels[gt_inds[hit] - 1]
        overlaps = torch.where(hit, iou.gather(1, (gt_inds - 1).clamp(min=0)[:, None])[:, 0], iou.new_zeros(()))
        return AssignResult(G, gt_inds, overlaps, labels=labels)


@register(BBOX_ASSIGNERS)
class HeuristicAssigner3D:
    """hungarian_assigner.py:49-91: every ground-truth box goes to its nearest proposal in the BEV plane (proposals of another
    class pushed ``dist_thre`` away when ``query_labels`` is given), kept if within ``dist_thre`` metres; a proposal claimed
    by several boxes keeps the nearest (the earliest box on a tie, as the reference's ascending loop with its strict ``<``).
    No shipped config selects it (FD:1077 even compares the type string with 'HeuristicAssigner', which is not the registered
    name); built for the registry's completeness.  One pass of device ops - no per-box host loop, no synchronisation:
    the reference's sequential update is the arg-min over (distance rank, box index) per proposal, a scatter-reduce."""

    def __init__(self, dist_thre=100, iou_calculator=dict(type='BboxOverlaps3D')):
        self.dist_thre = dist_thre                      # metres
        self.iou_calculator = build_iou_calculator(iou_calculator)

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None, query_labels=None):
        G, P = gt_bboxes.size(0), bboxes.size(0)
        labels = bboxes.new_full((P,), -1, dtype=torch.long)
        if P == 0 or G == 0:
            return AssignResult(G, bboxes.new_zeros((P,), dtype=torch.long), bboxes.new_zeros((P,)), labels=labels)
        dist = torch.norm(bboxes[:, 0:2][None, :, :] - gt_bboxes[:, 0:2][:, None, :], dim=-1)          # (G, P)
        if query_labels is not None:
            dist = dist + (query_labels[None] != gt_labels[:, None]) * self.dist_thre
        near_val, near_idx = dist.min(1)                                                               # per box
        order = torch.argsort(near_val, stable=True)            # rank 0 = the closest pair; equal distances: lower box index first
        rank = torch.empty_like(order)
        rank[order] = torch.arange(G, device=order.device)
        key = torch.where(near_val <= self.dist_thre, rank, torch.full_like(rank, G))
        best = torch.full((P,), G, dtype=torch.long, device=bboxes.device).scatter_reduce_(0, near_idx, key, 'amin')
        hit = best < G
        box_of = order[best.clamp(max=G - 1)]                                                          # (P,) winning box per proposal
        gt_inds = torch.where(hit, box_of + 1, torch.zeros_like(box_of))
        labels = torch.where(hit, gt_labels[box_of].long(), labels)
        iou = self.iou_calculator(bboxes, gt_bboxes)                                                   # (P, G)
        overlaps = torch.where(hit, iou.gather(1, box_of[:, None])[:, 0], iou.new_zeros(()))
        return AssignResult(G, gt_inds, overlaps, labels=labels)


# ------------------------------------------------------------------------------------------------ losses (mmdet 2.14)
def clip_sigmoid(x, eps=1e-4):
    """mmdet3d ``clip_sigmoid``."""
    return torch.clamp(x.sigmoid(), min=eps, max=1 - eps)


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == 'mean' else loss.sum() if reduction == 'sum' else loss
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


@register(LOSSES)
class FocalLoss(torch.nn.Module):
    """mmdet ``FocalLoss(use_sigmoid=True)``; integer targets, ``num_classes`` = background."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        if not use_sigmoid:
            raise NotImplementedError('Only sigmoid focal loss supported now.')
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        k = pred.size(1)
        t = F.one_hot(target, num_classes=k + 1)[:, :k].type_as(pred)
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction='none') * (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        if weight is not None and weight.shape != loss.shape:
            weight = weight.view(-1, 1) if weight.size(0) == loss.size(0) else weight.view(loss.size(0), -1)
        return self.loss_weight * weight_reduce_loss(loss, None if weight is None else weight.float(),
                                                     reduction_override or self.reduction, avg_factor)


@register(LOSSES)
class L1Loss(torch.nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        return self.loss_weight * weight_reduce_loss((pred - target).abs(), weight, reduction_override or self.reduction, avg_factor)


@register(LOSSES)
class GaussianFocalLoss(torch.nn.Module):
    def __init__(self, alpha=2.0, gamma=4.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.alpha, self.gamma, self.reduction, self.loss_weight = alpha, gamma, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        eps = 1e-12
        pos = -(pred + eps).log() * (1 - pred).pow(self.alpha) * target.eq(1)
        neg = -(1 - pred + eps).log() * pred.pow(self.alpha) * (1 - target).pow(self.gamma)
        return self.loss_weight * weight_reduce_loss(pos + neg, weight, reduction_override or self.reduction, avg_factor)


# ------------------------------------------------------------------------------------------------ head targets + loss
def _cfg_get(cfg, key, default=None):
    return cfg.get(key, default) if hasattr(cfg, 'get') else getattr(cfg, key, default)


def _box_tensor(b):
    return b.tensor if hasattr(b, 'tensor') else b


def head_get_targets_single(head, gt_bboxes_3d, gt_labels_3d, preds_dict, batch_idx):
    """FD:1022-1164 for one sample; ``gt_bboxes_3d`` a LiDAR box object (``.tensor``) or the (m, 7|9) tensor itself."""
    num_proposals = preds_dict['center'].shape[-1]
    score = preds_dict['heatmap'].detach().clone()
    vel = preds_dict['vel'].detach().clone() if 'vel' in preds_dict else None
    boxes_dict = head.bbox_coder.decode(score, preds_dict['rot'].detach().clone(), preds_dict['dim'].detach().clone(),
                                        preds_dict['center'].detach().clone(), preds_dict['height'].detach().clone(), vel)
    bboxes = boxes_dict[0]['bboxes']
    gt = _box_tensor(gt_bboxes_3d).to(score.device).float().contiguous()
    gt_labels_3d = gt_labels_3d.to(score.device)
    tc = head.train_cfg
    results = []
    for l in range(head.num_decoder_layers):
        sl = slice(head.num_proposals * l, head.num_proposals * (l + 1))
        layer_boxes = bboxes[sl].contiguous()
        if _cfg_get(tc['assigner'], 'type') != 'HungarianAssigner3D':
            raise NotImplementedError("train_cfg.assigner.type: only 'HungarianAssigner3D' (every shipped config)")
        res = head.bbox_assigner.assign(layer_boxes, gt, gt_labels_3d, score[..., sl], tc)
        if head.gt_center_limit is not None and res.max_overlaps is not None:                       # FD:1076-1080
            pos = res.gt_inds > 0
            far = (gt[res.gt_inds[pos] - 1][:, :2] - layer_boxes[pos][:, :2]).norm(dim=1) > head.gt_center_limit
            res.gt_inds[torch.nonzero(pos)[:, 0][far]] = 0
        results.append(res)
    ens = AssignResult(sum(r.num_gts for r in results), torch.cat([r.gt_inds for r in results]),
                       torch.cat([r.max_overlaps if r.max_overlaps is not None else bboxes.new_zeros(head.num_proposals)
                                  for r in results]),
                       labels=torch.cat([r.labels for r in results]))
    sampling = head.bbox_sampler.sample(ens, bboxes, gt)
    pos_inds, neg_inds = sampling.pos_inds, sampling.neg_inds
    assert len(pos_inds) + len(neg_inds) == num_proposals
    code = head.bbox_coder.code_size
    bbox_targets = bboxes.new_zeros(num_proposals, code)
    bbox_weights = bboxes.new_zeros(num_proposals, code)
    ious = ens.max_overlaps.clamp(0.0, 1.0)
    labels = bboxes.new_zeros(num_proposals, dtype=torch.long) + head.num_classes
    label_weights = bboxes.new_zeros(num_proposals, dtype=torch.long)
    if len(pos_inds) > 0:
        bbox_targets[pos_inds] = head.bbox_coder.encode(sampling.pos_gt_bboxes)
        bbox_weights[pos_inds] = 1.0
        labels[pos_inds] = gt_labels_3d[sampling.pos_assigned_gt_inds]
        pw = _cfg_get(tc, 'pos_weight')
        label_weights[pos_inds] = 1 if pw <= 0 else pw
    if len(neg_inds) > 0:
        label_weights[neg_inds] = 1
    grid, osf = _cfg_get(tc, 'grid_size'), _cfg_get(tc, 'out_size_factor')
    vox, pcr = _cfg_get(tc, 'voxel_size'), _cfg_get(tc, 'point_cloud_range')
    heatmap = ops.gaussian_heatmap_targets(gt, gt_labels_3d.long().contiguous(), head.num_classes, grid[1] // osf, grid[0] // osf,
                                           (osf, vox[0], vox[1], pcr[0], pcr[1]), _cfg_get(tc, 'gaussian_overlap'),
                                           _cfg_get(tc, 'min_radius'))
    mean_iou = ious[pos_inds].sum() / max(len(pos_inds), 1)
    return (labels[None], label_weights[None], bbox_targets[None], bbox_weights[None], ious[None], int(pos_inds.shape[0]),
            float(mean_iou), heatmap[None])


def _batched_targets_ok(head):
    a = getattr(head, 'bbox_assigner', None)
    return isinstance(a, HungarianAssigner3D) and isinstance(a.iou_calculator, BboxOverlaps3D) and \
        isinstance(a.iou_cost, IoU3DCost) and isinstance(a.cls_cost, FocalLossCost) and \
        isinstance(a.reg_cost, (BBoxBEVL1Cost, BBox3DL1Cost)) and isinstance(head.bbox_sampler, PseudoSampler)


def head_get_targets_batched(head, gt_bboxes_3d, gt_labels_3d, preds_dict):
    """FD:994-1164 for the whole batch with two host round trips instead of ~6 per frame and decoder stage: every cost matrix is
    built on the device, ONE copy carries them to the host for scipy's ``linear_sum_assignment`` (the reference's host step,
    hungarian_assigner.py:144-151), one copy brings the assignment back, and labels / weights / box targets / IoUs follow from
    masks and gathers.  Same values as ``head_get_targets_single`` per frame (tests compare both with the reference's)."""
    p = preds_dict[0]
    score = p['heatmap'].detach()
    B, K, N = score.shape
    L, P = head.num_decoder_layers, head.num_proposals
    assert N == L * P
    dev, tc, asg = score.device, head.train_cfg, head.bbox_assigner
    vel = p['vel'].detach().clone() if 'vel' in p else None
    boxes = head.bbox_coder.decode_all(score.clone(), p['rot'].detach().clone(), p['dim'].detach().clone(),
                                       p['center'].detach().clone(), p['height'].detach().clone(), vel)       # (B, N, 7|9)
    gts = [_box_tensor(g).to(dev).float().contiguous() for g in gt_bboxes_3d]
    gls = [l.to(dev).long() for l in gt_labels_3d]
    # ---- cost matrices (N, n_b) on the device; the rows of decoder stage l are [l*P, (l+1)*P)
    costs, ious = [], []
    for b in range(B):
        if gts[b].shape[0] == 0:
            costs.append(None), ious.append(None)
            continue
        cost, iou = asg.cost_matrix(boxes[b], gts[b], gls[b], score[b].T, tc)
        costs.append(cost)
        ious.append(iou)
    live = [c.reshape(-1) for c in costs if c is not None]
    flat = torch.cat(live).cpu().numpy() if live else np.zeros(0, np.float32)                    # host round trip 1
    gt_inds_h = np.zeros((B, N), np.int64)
    off = 0
    for b in range(B):
        if costs[b] is None:
            continue
        n = gts[b].shape[0]
        c = flat[off:off + N * n].reshape(N, n)
        off += N * n
        for l in range(L):
            gt_inds_h[b, l * P:(l + 1) * P] = asg.match(c[l * P:(l + 1) * P])
    gt_inds = torch.from_numpy(gt_inds_h).to(dev)
    # ---- targets from masks / gathers
    code = head.bbox_coder.code_size
    G = max(max(g.shape[0] for g in gts), 1)
    gt_pad = torch.stack([F.pad(g, (0, 0, 0, G - g.shape[0])) for g in gts])                      # (B, G, 7|9)
    gl_pad = torch.stack([F.pad(l, (0, G - l.shape[0])) for l in gls])
    idx = (gt_inds - 1).clamp(min=0)
    assigned = gt_inds > 0
    gt_of = gt_pad.gather(1, idx[..., None].expand(-1, -1, gt_pad.shape[-1]))                     # (B, N, 7|9)
    max_overlaps = torch.stack([
        torch.where(assigned[b], ious[b].gather(1, idx[b][:, None])[:, 0], ious[b].new_zeros(()))
        if ious[b] is not None else boxes.new_zeros(N) for b in range(B)])
    if head.gt_center_limit is not None:                                                          # FD:1076-1080
        far = (gt_of[..., :2] - boxes[..., :2]).norm(dim=-1) > head.gt_center_limit
        assigned = assigned & ~far                    # back to the negatives; their IoU stays in max_overlaps, as in the reference
    pos = assigned
    labels = torch.where(pos, gl_pad.gather(1, idx), torch.full_like(idx, head.num_classes))
    pw = _cfg_get(tc, 'pos_weight')
    label_weights = torch.where(pos, torch.full_like(idx, 1 if pw <= 0 else int(pw)), torch.ones_like(idx))
    enc = head.bbox_coder.encode(gt_of.reshape(B * N, -1)).view(B, N, code)
    bbox_targets = torch.where(pos[..., None], enc, enc.new_zeros(()))
    bbox_weights = pos[..., None].expand(-1, -1, code).to(enc.dtype)
    iou_t = max_overlaps.clamp(0.0, 1.0)
    n_pos = pos.sum(1)
    mean_iou = (iou_t * pos).sum(1) / n_pos.clamp(min=1)
    grid, osf = _cfg_get(tc, 'grid_size'), _cfg_get(tc, 'out_size_factor')
    vox, pcr = _cfg_get(tc, 'voxel_size'), _cfg_get(tc, 'point_cloud_range')
    heatmap = torch.stack([ops.gaussian_heatmap_targets(gts[b], gls[b].contiguous(), head.num_classes, grid[1] // osf,
                                                        grid[0] // osf, (osf, vox[0], vox[1], pcr[0], pcr[1]),
                                                        _cfg_get(tc, 'gaussian_overlap'), _cfg_get(tc, 'min_radius'))
                           for b in range(B)])
    scal = torch.cat([n_pos.sum()[None].to(mean_iou.dtype), mean_iou.mean()[None]]).tolist()      # host round trip 2
    return labels, label_weights, bbox_targets, bbox_weights, iou_t, int(round(scal[0])), float(scal[1]), heatmap


def head_get_targets(head, gt_bboxes_3d, gt_labels_3d, preds_dict):
    """FD:994-1020: ``preds_dict`` = the [dict] of one output level."""
    if _batched_targets_ok(head) and getattr(head, 'batched_targets', True):
        return head_get_targets_batched(head, gt_bboxes_3d, gt_labels_3d, preds_dict)
    res = []
    for b in range(len(gt_bboxes_3d)):
        one = {k: v[b:b + 1] for k, v in preds_dict[0].items() if torch.is_tensor(v)}
        res.append(head_get_targets_single(head, gt_bboxes_3d[b], gt_labels_3d[b], one, b))
    cat = lambda i: torch.cat([r[i] for r in res], 0)                                    # noqa: E731
    return (cat(0), cat(1), cat(2), cat(3), cat(4), int(np.sum([r[5] for r in res])), float(np.mean([r[6] for r in res])),
            cat(7))


def head_loss(head, gt_bboxes_3d, gt_labels_3d, preds_dicts, **kwargs):
    """FD:1166-1311 (without the heatmap_box branch, which needs mmdet3d's DCNSeparateHead; no shipped config enables it)."""
    labels, label_weights, bbox_targets, bbox_weights, ious, num_pos, matched_ious, heatmap = head_get_targets(
        head, gt_bboxes_3d, gt_labels_3d, preds_dicts[0])
    p = dict(preds_dicts[0][0])
    out = {}
    dense = p['dense_heatmap']
    if isinstance(dense, (tuple, list)):
        masks = torch.cat(p['multistage_masks'], 0) if 'multistage_masks' in p else None
        hm = heatmap.repeat(len(dense), 1, 1, 1)
        if masks is not None:
            hm = hm * masks
        loss_hm = head.loss_heatmap(clip_sigmoid(torch.cat(list(dense), 0)), hm, weight=masks,
                                    avg_factor=max(hm.eq(1).float().sum().item(), 1))
    else:
        loss_hm = head.loss_heatmap(clip_sigmoid(dense), heatmap, avg_factor=max(heatmap.eq(1).float().sum().item(), 1))
    out['loss_heatmap'] = loss_hm * head.loss_weight_heatmap
    code_weights = _cfg_get(head.train_cfg, 'code_weights', None)
    n = head.num_proposals
    for l in range(head.num_decoder_layers):
        sl = slice(l * n, (l + 1) * n)
        cls_score = p['heatmap'][..., sl].permute(0, 2, 1).reshape(-1, head.num_classes)
        out[f'layer_{l}_loss_cls'] = head.loss_cls(cls_score, labels[..., sl].reshape(-1), label_weights[..., sl].reshape(-1),
                                                   avg_factor=max(num_pos, 1))
        parts = [p[k][..., sl] for k in ('center', 'height', 'dim', 'rot')] + ([p['vel'][..., sl]] if 'vel' in p else [])
        pred = torch.cat(parts, 1).permute(0, 2, 1)
        w = bbox_weights[:, sl, :] * bbox_weights.new_tensor(code_weights)
        out[f'layer_{l}_loss_bbox'] = head.loss_bbox(pred, bbox_targets[:, sl, :], w, avg_factor=max(num_pos, 1))
    if head.add_gt_groups > 0 and 'batch_valid_gt_mask' not in p:
        # no frame of this rank's batch has a ground-truth box: the terms are zero, but the KEY SET of the loss dict must not
        # depend on the data (mmdet's _parse_losses all-reduces per key under DDP)
        zero = p['heatmap'].sum() * 0
        out['gt_query_loss_box'], out['gt_query_loss_cls'] = zero, zero.clone()
    if head.add_gt_groups > 0 and 'batch_valid_gt_mask' in p:                                          # FD:1222-1254
        nl = head.num_decoder_layers
        valid = p['batch_valid_gt_mask'].float()
        q_labels = p['batch_gt_query_labels'].repeat(1, nl)
        pq = torch.cat([p['center_gtgroups'], p['height_gtgroups'], p['rot_gtgroups'], p['dim_gtgroups']]
                       + ([p['vel_gtgroups']] if 'vel' in p else []), 1).permute(0, 2, 1)
        sq = p['heatmap_gtgroups'].permute(0, 2, 1).reshape(-1, head.num_classes)
        tg = torch.zeros((pq.shape[0], head.max_num_gts, head.bbox_coder.code_size), dtype=pq.dtype, device=pq.device)
        for b, g in enumerate(gt_bboxes_3d):
            t = _box_tensor(g)
            tg[b, :len(t)] = head.bbox_coder.encode(t.to(pq.device))
        positive = q_labels != head.num_classes
        tg = tg.repeat(1, head.add_gt_groups * nl, 1)
        rw = valid[:, :, None].repeat(1, nl, tg.shape[-1]) * valid.new_tensor(code_weights) * positive[..., None].float()
        af = max(sum(head.num_gts) * head.add_gt_groups * nl, 1)
        out['gt_query_loss_box'] = head.loss_bbox(pq, tg, rw, avg_factor=af) * head.gt_query_loss_weight
        out['gt_query_loss_cls'] = head.loss_cls(sq, q_labels.reshape(-1), valid.repeat(1, nl).reshape(-1),
                                                 avg_factor=af) * head.gt_query_loss_weight
    out['matched_ious'] = out['layer_0_loss_cls'].new_tensor(matched_ious)
    return out
