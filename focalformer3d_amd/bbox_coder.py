"""``TransFusionBBoxCoder`` - registry-compatible mirror of the reference coder
(projects/mmdet3d_plugin/core/bbox/coders/transfusion_bbox_coder.py:7-158) on the HIP path."""
import torch

from . import ops
from .registry import BBOX_CODERS, register


@register(BBOX_CODERS)
class TransFusionBBoxCoder:
    def __init__(self, pc_range, out_size_factor, voxel_size, post_center_range=None, score_threshold=None,
                 code_size=8):
        self.pc_range = pc_range
        self.out_size_factor = out_size_factor
        self.voxel_size = voxel_size
        self.post_center_range = post_center_range
        self.score_threshold = score_threshold
        self.code_size = code_size

    @property
    def coder_params(self):
        """(out_size_factor, voxel_x, voxel_y, pc_x, pc_y) as the C ABI takes them."""
        return (float(self.out_size_factor), float(self.voxel_size[0]), float(self.voxel_size[1]),
                float(self.pc_range[0]), float(self.pc_range[1]))

    def encode(self, dst_boxes):
        """BC:24-37 (training targets; plain tensor arithmetic, works on any device)."""
        t = torch.zeros([dst_boxes.shape[0], self.code_size], device=dst_boxes.device)
        t[:, 0] = (dst_boxes[:, 0] - self.pc_range[0]) / (self.out_size_factor * self.voxel_size[0])
        t[:, 1] = (dst_boxes[:, 1] - self.pc_range[1]) / (self.out_size_factor * self.voxel_size[1])
        t[:, 3:6] = (dst_boxes[:, 3:6] + 1e-6).log()
        t[:, 2] = dst_boxes[:, 2] + dst_boxes[:, 5] * 0.5
        t[:, 6] = torch.sin(dst_boxes[:, 6])
        t[:, 7] = torch.cos(dst_boxes[:, 6])
        if self.code_size == 10:
            t[:, 8:10] = dst_boxes[:, 7:]
        return t

    def decode_padded(self, heatmap, rot, dim, center, height, vel, qscore=None, qlabel=None, max_out=None, filter=True):
        """BC:71-158 ``decode`` on the device, without the data-dependent compaction:
        returns padded (boxes (B,n,7|9), scores (B,n), labels int32 (B,n), count int32 (B,)).
        ``heatmap`` is the already-fused score (B,K,N) unless ``qscore``/``qlabel`` are given, in which
        case it is the raw class logits and FD:1317-1321 is fused in as well.  ``filter=False``: BC's unfiltered branch - every
        query in its own row, count = N (no range test, so a NaN / inf box neither vanishes nor shifts the rows behind it)."""
        if filter and self.post_center_range is None:
            raise NotImplementedError('Need to reorganize output as a batch, only support post_center_range '
                                      'is not None for now!')      # BC:155-158
        B, K, N = heatmap.shape
        preds = dict(heatmap=heatmap, center=center, height=height, dim=dim, rot=rot)
        if vel is not None:
            preds['vel'] = vel
        if qscore is None:
            # plain decode: score = max_c heatmap, label = argmax_c.  Expressed through the same kernel by
            # feeding logit(+inf) == 1 is not exact, so do the tiny max on the device with torch.
            scores, labels = heatmap.max(1)
            qscore = torch.zeros_like(heatmap).scatter_(1, labels[:, None], scores[:, None])
            qlabel = labels
            preds['heatmap'] = torch.full_like(heatmap, 1e4)        # sigmoid(1e4) == 1.0 exactly in fp32
        return ops.box_decode({k: v.contiguous() for k, v in preds.items()}, 0, N, qscore.contiguous(),
                              qlabel.contiguous(), self.coder_params, self.post_center_range if filter else None,
                              (self.score_threshold or 0.0) if filter else 0.0, (max_out or N) if filter else N)

    def decode_all(self, heatmap, rot, dim, center, height, vel):
        """BC:71-158 ``decode(filter=False)`` for the whole batch without the per-sample lists (and without their host
        synchronisation): boxes (B, N, 7|9), row q = query q."""
        return self.decode_padded(heatmap, rot, dim, center, height, vel, filter=False)[0]

    def decode(self, heatmap, rot, dim, center, height, vel, filter=False):
        """BC:71-158: list of dict(bboxes, scores, labels) per sample."""
        boxes, scores, labels, count = self.decode_padded(heatmap, rot, dim, center, height, vel, filter=filter)
        counts = count.tolist()
        return [dict(bboxes=boxes[i, :n], scores=scores[i, :n], labels=labels[i, :n].long())
                for i, n in enumerate(counts)]
