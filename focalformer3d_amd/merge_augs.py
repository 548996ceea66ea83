"""Test-time-augmentation box merging on MI355X - the compute of ``merge_aug_bboxes_3d``
(projects/mmdet3d_plugin/core/post_processing/merge_augs.py:13-184).

What is mirrored: mapping the per-augmentation detections back (mmdet3d ``bbox3d_mapping_back``: undo flips, then the
scale), per-class rotated-IoU NMS at 0.1 (``nms_gpu``), IoU-weighted box voting at 0.65 against all boxes of the class
(``boxes_iou_bev``; yaw as a circular mean), and the final top-500 by score - with the reference's hard-wired constants
(merge_augs.py:118-123, 153-154).  Both IoU ops are HIP kernels behind the C ABI (``ff3d_nms_bev``, ``ff3d_boxes_iou_bev``).
Not mirrored: the reference's pickle side channel (``./merge_augs*/sampleidx_*.pkl``, the ``ensemble`` switch) - file
plumbing of its evaluation scripts, not arithmetic.
"""
import math

import torch

from . import ops

NMS_THR, VOTE_IOU_THRESH, MAX_NUM = 0.1, 0.65, 500


def xywhr2xyxyr(bev):
    """mmdet3d ``xywhr2xyxyr``: (x, y, w, l, r) -> (x - w/2, y - l/2, x + w/2, y + l/2, r)."""
    out = torch.empty_like(bev)
    hw, hl = bev[:, 2] / 2, bev[:, 3] / 2
    out[:, 0], out[:, 1], out[:, 2], out[:, 3], out[:, 4] = bev[:, 0] - hw, bev[:, 1] - hl, bev[:, 0] + hw, bev[:, 1] + hl, bev[:, 4]
    return out


def bev_of(boxes):
    """``LiDARInstance3DBoxes.bev`` (mmdet3d 0.17.1): columns (x, y, x_size, y_size, yaw)."""
    return boxes[:, [0, 1, 3, 4, 6]]


def bbox3d_mapping_back(boxes, scale_factor, flip_horizontal, flip_vertical):
    """mmdet3d 0.17.1 ``bbox3d_mapping_back`` for LiDAR boxes (x, y, z, w, l, h, yaw[, vx, vy]): flip 'horizontal' negates
    y and maps yaw -> -yaw + pi; flip 'vertical' negates x and maps yaw -> -yaw; then every metric column (incl. velocity)
    is scaled by 1 / scale_factor."""
    b = boxes.clone()
    if flip_horizontal:
        b[:, 1] = -b[:, 1]
        b[:, 6] = -b[:, 6] + math.pi
        if b.shape[1] > 7:
            b[:, 8] = -b[:, 8]
    if flip_vertical:
        b[:, 0] = -b[:, 0]
        b[:, 6] = -b[:, 6]
        if b.shape[1] > 7:
            b[:, 7] = -b[:, 7]
    s = 1.0 / scale_factor
    b[:, :6] *= s
    b[:, 7:] *= s
    return b


def merge_boxes(aug_boxes, aug_scores, aug_labels):
    """merge_augs.py:125-184 on the concatenated, mapped-back detections: (n, box_dim) fp32, (n,), (n,) int ->
    (boxes, scores, labels) sorted by score, at most 500."""
    if not aug_boxes.is_cuda:
        raise RuntimeError('merge_boxes: inputs must live on the MI355X (HIP) device - no CPU fallback')
    if aug_labels.numel() == 0:
        return aug_boxes, aug_scores, aug_labels
    for_nms = xywhr2xyxyr(bev_of(aug_boxes)).contiguous()
    out_b, out_s, out_l = [], [], []
    for cls in range(int(aug_labels.max().item()) + 1):
        sel = aug_labels == cls
        if not bool(sel.any()):
            continue
        boxes_i, nms_i, scores_i, labels_i = aug_boxes[sel], for_nms[sel].contiguous(), aug_scores[sel].contiguous(), aug_labels[sel]
        keep = ops.nms_bev(nms_i, scores_i, NMS_THR)
        chosen = boxes_i[keep]
        iou = ops.boxes_iou_bev(xywhr2xyxyr(bev_of(chosen)).contiguous(), nms_i)
        iou = torch.where(iou < VOTE_IOU_THRESH, torch.zeros_like(iou), iou)
        wsum = iou.sum(dim=1)
        voted = (iou[:, :, None] * boxes_i[None]).sum(dim=1) / (wsum[:, None] + 1e-6)
        voted[:, 6] = torch.atan2((iou * torch.sin(boxes_i[None, :, 6])).sum(dim=1) / (wsum + 1e-6),
                                  (iou * torch.cos(boxes_i[None, :, 6])).sum(dim=1) / (wsum + 1e-6))
        out_b.append(voted)
        out_s.append(scores_i[keep])
        out_l.append(labels_i[keep])
    boxes, scores, labels = torch.cat(out_b), torch.cat(out_s), torch.cat(out_l)
    order = scores.sort(0, descending=True)[1][:min(MAX_NUM, aug_boxes.shape[0])]
    return boxes[order], scores[order], labels[order]


def merge_aug_bboxes_3d(aug_results, img_metas, test_cfg=None):
    """merge_augs.py:13 ``merge_aug_bboxes_3d(aug_results, img_metas, test_cfg)``: aug_results = list of dicts with
    ``boxes_3d`` (box object with ``.tensor`` or a plain (n, box_dim) tensor), ``scores_3d``, ``labels_3d``; img_metas =
    per augmentation ``[{'pcd_scale_factor', 'pcd_horizontal_flip', 'pcd_vertical_flip', 'box_type_3d'?}]``.  Returns the
    reference's result dict (tensors on the host, as ``bbox3d2result``)."""
    assert len(aug_results) == len(img_metas), \
        f'"aug_results" should have the same length as "img_metas", got {len(aug_results)} and {len(img_metas)}'
    rec_b, rec_s, rec_l = [], [], []
    for res, info in zip(aug_results, img_metas):
        m = info[0]
        t = res['boxes_3d']
        t = t.tensor if hasattr(t, 'tensor') else t
        rec_b.append(bbox3d_mapping_back(t, m['pcd_scale_factor'], m['pcd_horizontal_flip'], m['pcd_vertical_flip']))
        rec_s.append(res['scores_3d'])
        rec_l.append(res['labels_3d'])
    boxes, scores, labels = merge_boxes(torch.cat(rec_b).contiguous(), torch.cat(rec_s).contiguous(), torch.cat(rec_l))
    box_type = img_metas[0][0].get('box_type_3d')
    boxes = boxes.cpu()
    if box_type is not None:
        boxes = box_type(boxes, box_dim=boxes.shape[-1])
    return dict(boxes_3d=boxes, scores_3d=scores.cpu(), labels_3d=labels.cpu())
