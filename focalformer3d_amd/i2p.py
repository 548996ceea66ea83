"""``I2P`` - the LiDAR<->camera projection feature sampler on MI355X.

Mirror of projects/mmdet3d_plugin/models/utils/encoder_utils.py:184-261 (same constructor, same
``forward(lidar_feat, img_feat, img_metas)``, same parameter names: ``learnedAlign.*``).

The reference materialises the (Ncam, C, Z*H*W) sampled tensor (~1 GB at the BASELINE shape) and runs a
1-head nn.MultiheadAttention over the Z height samples of every pillar.  Here the attention is folded
around one fused gfx950 kernel (``ff3d_cam_sample``: projection + bilinear gather + masked multi-view mean +
online softmax over Z), so only two small GEMMs remain:
    qk  = lidar_feat @ (Wk^T Wq / sqrt(C))^T + Wk^T bq / sqrt(C)        (the key bias is softmax-invariant)
    out = (Wo Wv) ctx + (Wo bv + bo),   zero where no height sample is visible in any camera (EU:256-258)
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .coord_transform import fold_into_lidar2img

# the projections around the camera sampler on the own split-fp16 linear kernel (FF3D_I2P_OWN_LINEAR=0: hipBLASLt fp32)
OWN_LINEAR = os.environ.get('FF3D_I2P_OWN_LINEAR', '1') != '0'
_PC_RANGE = (-54.0, -54.0, -5.0, 54.0, 54.0, 3.0)    # hard-coded in the reference, EU:210


class I2P(nn.Module):
    def __init__(self, pts_channels, img_channels, dropout, max_points_height=5):
        super().__init__()
        self.pts_channels = pts_channels
        self.img_channels = img_channels
        self.dropout = dropout
        self.max_points_height = max_points_height
        self.learnedAlign = nn.MultiheadAttention(pts_channels, 1, dropout=dropout, kdim=img_channels,
                                                  vdim=img_channels, batch_first=True)
        self._folded = None
        self.register_load_state_dict_post_hook(lambda m, k: m.invalidate_cache())

    def invalidate_cache(self):
        self._folded = None

    def train(self, mode=True):
        self._folded = None
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._folded = None
        return super()._apply(fn, *a, **k)

    def _fold(self):
        from .layers import weight_signature
        sig = weight_signature(self.learnedAlign.parameters())
        if self._folded is None or self._fold_sig != sig:
            self._fold_sig = sig
            a, C = self.learnedAlign, self.pts_channels
            with torch.no_grad():
                if a.in_proj_weight is not None:
                    wq, wk, wv = a.in_proj_weight.chunk(3, 0)
                else:
                    wq, wk, wv = a.q_proj_weight, a.k_proj_weight, a.v_proj_weight
                bq, _, bv = a.in_proj_bias.chunk(3, 0)
                s = 1.0 / float(C) ** 0.5
                self._folded = ((wk.t() @ wq * s).contiguous(), (wk.t() @ bq * s).contiguous(),
                                (a.out_proj.weight @ wv).contiguous(),
                                (a.out_proj.weight @ bv + a.out_proj.bias).contiguous())
        return self._folded

    def _forward_train(self, lidar_feat, img_feat, img_metas):
        """EU:194-261 under autograd (training mode: gradients reach the LiDAR map, the camera maps and ``learnedAlign``;
        attention dropout active).  The differentiable route keeps the reference's formulation - projection (no gradient),
        ``F.grid_sample`` of every camera map at the projected pillar points, masked multi-view mean, one-head
        ``nn.MultiheadAttention`` over the height samples of every visible pillar - on the framework's ops; the fused
        ``ff3d_cam_sample`` kernel is the inference path."""
        import torch.nn.functional as F
        from .coord_transform import apply_3d_transformation
        B, C, H, W = lidar_feat.shape
        Z, dev = self.max_points_height, lidar_feat.device
        out = torch.zeros_like(lidar_feat)
        lo = lidar_feat.new_tensor(_PC_RANGE[:3])
        span = lidar_feat.new_tensor(_PC_RANGE[3:]) - lo
        zz, yy, xx = torch.meshgrid(torch.arange(Z, device=dev), torch.arange(H, device=dev), torch.arange(W, device=dev),
                                    indexing='ij')                                       # flat index (z*H + y)*W + x, EU:174-182
        grid = torch.stack([xx, yy, zz], -1).reshape(-1, 3).float() + 0.5
        grid = grid / lidar_feat.new_tensor([W, H, Z]) * span + lo
        for b, meta in enumerate(img_metas):
            with torch.no_grad():
                pts = apply_3d_transformation(grid, 'LIDAR', meta, reverse=True) if meta.get('transformation_3d_flow') else grid
                l2i = torch.as_tensor(np.asarray(meta['lidar2img'], dtype=np.float32), device=dev)
                cam = torch.matmul(l2i[:, None], torch.cat([pts, torch.ones_like(pts[:, :1])], -1)[None, :, :, None]).squeeze(-1)
                mask = cam[..., 2:3] > 1e-5
                xy = cam[..., :2] / cam[..., 2:3].clamp_min(1e-5)
                if 'img_aug_matrix' in meta:                                                # EU:230-233
                    aug = torch.as_tensor(meta['img_aug_matrix'], dtype=torch.float32, device=dev)
                    xy1 = torch.cat([xy, torch.ones_like(xy[..., :1])], -1)
                    xy = (aug[:, None, :3, :3].matmul(xy1.unsqueeze(-1)).squeeze(-1) + aug[:, None, :3, 3])[..., :2]
                ih, iw = meta['input_shape'][:2]
                xy = (torch.stack([xy[..., 0] / iw, xy[..., 1] / ih], -1) - 0.5) * 2
                mask = (mask & (xy[..., 0:1] > -1.0) & (xy[..., 0:1] < 1.0) & (xy[..., 1:2] > -1.0) & (xy[..., 1:2] < 1.0))[..., 0]
            ncam = xy.shape[0]
            sampled = F.grid_sample(img_feat[b], xy.unsqueeze(-2), mode='bilinear', padding_mode='zeros',
                                    align_corners=False).squeeze(-1).view(ncam, -1, Z, H, W)
            m = mask.view(ncam, 1, Z, H, W).to(sampled.dtype)
            red = ((sampled * m).sum(0) / (m.sum(0) + 1e-10)).flatten(2, 3).transpose(0, 2)       # (HW, Z, Ci)
            kmask = (m[:, 0].sum(0) > 0).view(Z, H * W).t()                                       # (HW, Z)
            valid = kmask.any(1)
            q = lidar_feat[b].flatten(1, 2).t().unsqueeze(1)                                      # (HW, 1, C)
            attn = lidar_feat.new_zeros(H * W, 1, C)
            if bool(valid.any()):
                attn[valid] = self.learnedAlign(q[valid], red[valid], red[valid], attn_mask=(~kmask[valid])[:, None, :])[0]
            out[b] = attn.squeeze(1).t().reshape(C, H, W)
        return out

    def forward(self, lidar_feat, img_feat, img_metas, **kwargs):
        """lidar_feat (B,C,H,W); img_feat (B,Ncam,Ci,Hi,Wi) NCHW camera maps; img_metas: per-sample dicts
        with 'lidar2img' (Ncam,4,4), 'input_shape' (h,w) and optionally 'img_aug_matrix' (Ncam,4,4)."""
        if not lidar_feat.is_cuda:
            raise RuntimeError('I2P: inputs must live on the MI355X (HIP) device - no CPU fallback')
        if self.training:
            return self._forward_train(lidar_feat, img_feat, img_metas)
        B, C, H, W = lidar_feat.shape
        _, ncam, Ci, Hi, Wi = img_feat.shape
        dev = lidar_feat.device
        with torch.no_grad():
            # EU:222: pillar points live in the (possibly augmented / flipped) LiDAR frame of the BEV map; the recorded flow is
            # undone before projecting - one affine map per frame, folded into lidar2img (coord_transform.py)
            l2i_host = np.ascontiguousarray(np.asarray([
                fold_into_lidar2img(m['lidar2img'], m) if m.get('transformation_3d_flow')
                else np.asarray(m['lidar2img'], dtype=np.float32) for m in img_metas], dtype=np.float32))
            aug_host = None
            if 'img_aug_matrix' in img_metas[0]:
                aug_host = np.ascontiguousarray(np.asarray([np.asarray(m['img_aug_matrix'], dtype=np.float32) for m in img_metas]))
            # the camera matrices on the device, re-uploaded only when their VALUES change: a serving loop over one rig (and a
            # captured graph of neck + head, round 5) uploads them once - no host-to-device copy inside the step
            key = (l2i_host.tobytes(), None if aug_host is None else aug_host.tobytes(), str(dev))
            cached = self.__dict__.get('_cam_dev')
            if cached is None or cached[0] != key:
                cached = (key, torch.from_numpy(l2i_host).to(dev).contiguous(),
                          None if aug_host is None else torch.from_numpy(aug_host).to(dev).contiguous())
                self.__dict__['_cam_dev'] = cached
            l2i, aug = cached[1], cached[2]
            wqk, bqk, wov, bov = self._fold()
            cl = img_feat.permute(0, 1, 3, 4, 2)
            if cl.is_contiguous():              # channels-last memory already (FocalEncoder's shared_conv_img writes it, round 5)
                img_cl = cl
            else:
                img_cl = ops.nchw_to_nhwc(img_feat.contiguous().view(B * ncam, Ci, Hi, Wi)).view(B, ncam, Hi, Wi, Ci)
            q_cl = ops.nchw_to_nhwc(lidar_feat.contiguous())                       # (B,H,W,C)
            own = OWN_LINEAR and C % 32 == 0 and Ci % 32 == 0
            if own:     # round 4: the two projections around the sampler on the split-fp16 linear kernel (were hipBLASLt fp32 GEMMs)
                if getattr(self, '_split_sig', None) is not self._folded:
                    self._split = (ops.split_weight_f16(wqk, bias=bqk), ops.split_weight_f16(wov, bias=bov))
                    self._split_sig = self._folded
                qk = ops.linear_f16x3(q_cl.view(B, H * W, C), self._split[0], bqk)
            else:
                ops.note_vendor('I2P query projection', B * H * W, wqk.shape[0], C)
                qk = torch.nn.functional.linear(q_cl.view(B, H * W, C), wqk, bqk)  # (B,HW,Ci)
            ctx, valid = ops.cam_sample(img_cl, l2i, aug, qk.contiguous(), H, W, self.max_points_height, _PC_RANGE,
                                        tuple(float(v) for v in img_metas[0]['input_shape'][:2]))
            if own:
                rows = ops.linear_f16x3(ctx, self._split[1], bov) * valid.view(B, H * W, 1).to(ctx.dtype)      # (B,HW,C)
                return ops.nchw_to_nhwc(rows.view(B, H * W, C, 1)).view(B, C, H, W)     # (B,HW,C) -> (B,C,HW): one transposing pass
            ops.note_vendor('I2P output projection', B * H * W, wov.shape[0], wov.shape[1])
            out = torch.matmul(wov, ctx.transpose(1, 2)) + bov[:, None]            # (B,C,HW)
            out = out * valid.view(B, 1, H * W).to(out.dtype)
            return out.view(B, C, H, W)
