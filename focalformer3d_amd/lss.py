"""``LiftSplatShoot`` - the Lift-Splat-Shoot camera->BEV branch of the neck on MI355X.

Mirror of projects/mmdet3d_plugin/models/necks/lss.py:125-383 (``CamEncode``, ``LiftSplatShoot``): same constructor,
``forward(x, rots, trans, ..., img_metas=...) -> (bev, depth)`` and parameter names (``frustum``,
``camencode.depthnet.*``, ``bevencode.*``).  Inference only, no point-cloud augmentation undo (test time).

MI355X design: the depth net runs as ONE GEMM on the NHWC camera maps whose output rows are [features | depth logits]
per pixel; the depth-weighted outer product of the reference (1.4 GB per frame, lss.py:135-141) is never materialised -
``ff3d_lss_cells`` turns the camera poses straight into one 4-byte BEV-cell key per frustum point (the reference's
chain of (B,N,D,H,W,3,3) batched matmuls is never built), the keys are radix-sorted (indices only) and the fused
``ff3d_lss_splat`` kernel reduces every cell directly from the L2-resident feature rows and depth probabilities, as exact
interval sums (the reference's default path uses an fp32 cumsum trick instead, lss.py:97-108; ``newbevpool`` its
bev_pool extension).  The BEV encoder's four 3x3 convs (843 GFLOP per frame, 85 % of the branch) run on the split-fp16
MFMA conv kernels with BatchNorm folded and shift + ReLU in the epilogue, chained through (hi, lo') NHWC pairs.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .coord_transform import apply_3d_transformation, fold_into_cam2ego, has_transformation


class CamEncode(nn.Module):
    """lss.py:125-148."""

    def __init__(self, D, C, inputC):
        super().__init__()
        self.D, self.C = D, C
        self.depthnet = nn.Conv2d(inputC, D + C, kernel_size=1, padding=0)


class LiftSplatShoot(nn.Module):
    def __init__(self, img_scale=(900, 1600), camera_depth_range=[4.0, 45.0, 1.0], pc_range=[-50, -50, -5, 50, 50, 3],
                 downsample=4, grid=3, inputC=256, outputC=128, camC=64, newbevpool=False):
        super().__init__()
        self.pc_range, self.img_scale, self.grid = pc_range, img_scale, grid
        self.grid_conf = {'xbound': [pc_range[0], pc_range[3], grid], 'ybound': [pc_range[1], pc_range[4], grid],
                          'zbound': [pc_range[2], pc_range[5], grid], 'dbound': camera_depth_range}
        rows = [self.grid_conf[k] for k in ('xbound', 'ybound', 'zbound')]
        self.dx = torch.Tensor([r[2] for r in rows])                                # lss.py:82-87 gen_dx_bx
        self.bx = torch.Tensor([r[0] + r[2] / 2.0 for r in rows])
        self.nx = torch.LongTensor([(r[1] - r[0]) / r[2] for r in rows])
        self.downsample = downsample
        self.fH, self.fW = img_scale[0] // downsample, img_scale[1] // downsample
        self.camC, self.inputC = camC, inputC
        self.frustum = self.create_frustum()
        self.D = self.frustum.shape[0]
        self.camencode = CamEncode(self.D, camC, inputC)
        self.newbevpool = newbevpool
        self.use_quickcumsum = True
        self.dense_mode = os.environ.get('FF3D_DENSE_MODE', 'f16x3')    # BEV-encoder convs: 'f16x3' (own split-fp16 MFMA kernels) | 'vendor'
        z = self.grid_conf['zbound']
        cz = int(camC * ((z[1] - z[0]) // z[2]))
        chans = [cz, cz, 512, 512, outputC]
        layers = []
        for i in range(4):
            layers += [nn.Conv2d(chans[i], chans[i + 1], kernel_size=3, padding=1, bias=False),
                       nn.BatchNorm2d(chans[i + 1]), nn.ReLU(inplace=True)]
        self.bevencode = nn.Sequential(*layers)

    def create_frustum(self):
        """lss.py:217-230: image-plane grid (x_px, y_px, depth) at feature resolution."""
        ogfH, ogfW = self.img_scale
        ds = torch.arange(*self.grid_conf['dbound'], dtype=torch.float).view(-1, 1, 1).expand(-1, self.fH, self.fW)
        D = ds.shape[0]
        xs = torch.linspace(0, ogfW - 1, self.fW, dtype=torch.float).view(1, 1, self.fW).expand(D, self.fH, self.fW)
        ys = torch.linspace(0, ogfH - 1, self.fH, dtype=torch.float).view(1, self.fH, 1).expand(D, self.fH, self.fW)
        return nn.Parameter(torch.stack((xs, ys, ds), -1), requires_grad=False)

    def get_geometry(self, rots, trans, post_rots=None, post_trans=None, extra_rots=None, extra_trans=None, img_metas=None):
        """lss.py:232-276 -> (B, N, D, fH, fW, 3) ego-frame points.  Kept for API parity / debugging: the forward path does
        not call it (``ff3d_lss_cells`` fuses this algebra with the binning and never stores the points)."""
        B, N, _ = trans.shape
        if img_metas is not None and 'img_aug_matrix' in img_metas[0]:
            aug = torch.stack([torch.as_tensor(m['img_aug_matrix'], dtype=torch.float32) for m in img_metas]).to(rots)
            post_rots, post_trans = aug[..., :3, :3], aug[..., :3, 3]
        if post_rots is not None or post_trans is not None:
            pts = self.frustum - post_trans.view(B, N, 1, 1, 1, 3)
            pts = torch.inverse(post_rots).view(B, N, 1, 1, 1, 3, 3).matmul(pts.unsqueeze(-1))
        else:
            pts = self.frustum.repeat(B, N, 1, 1, 1, 1).unsqueeze(-1)
        pts = torch.cat((pts[..., :2, :] * pts[..., 2:3, :], pts[..., 2:3, :]), 5)
        pts = rots.view(B, N, 1, 1, 1, 3, 3).matmul(pts).squeeze(-1) + trans.view(B, N, 1, 1, 1, 3)
        if has_transformation(img_metas):                                                   # lss.py:262-265
            pts = torch.stack([apply_3d_transformation(pts[b].reshape(-1, 3), 'LIDAR', img_metas[b], reverse=False)
                               .view(pts.shape[1:]) for b in range(B)])
        if extra_rots is not None:
            pts = extra_rots.view(B, N, 1, 1, 1, 3, 3).matmul(pts.unsqueeze(-1)).squeeze(-1)
        if extra_trans is not None:
            pts = pts + extra_trans.view(B, N, 1, 1, 1, 3)
        return pts

    def _aug(self, img_metas, post_rots, post_trans, ref):
        """Image-augmentation undo as (inverse rotation, translation) (B, N, 3, 3) / (B, N, 3), or (None, None)."""
        if img_metas is not None and 'img_aug_matrix' in img_metas[0]:                      # lss.py:236-240, host matrices
            import numpy as np
            aug = torch.from_numpy(np.asarray([np.asarray(m['img_aug_matrix'], dtype=np.float32) for m in img_metas]))
            return (torch.inverse(aug[..., :3, :3]).contiguous().to(ref.device),
                    aug[..., :3, 3].contiguous().to(ref.device))
        if post_rots is None and post_trans is None:
            return None, None
        return torch.inverse(post_rots).contiguous(), post_trans.contiguous()

    def cell_table(self, rots, trans, post_rots=None, post_trans=None, extra_rots=None, extra_trans=None, img_metas=None):
        """Frustum points binned and ordered by BEV cell: (src (E_total) int32 entry ids sorted by cell key, offsets
        (n_cells + 1) int32).  Geometry + binning run in ``ff3d_lss_cells`` (lss.py:232-276, :324-337); the ordering is a
        stable radix sort of the 4-byte keys (rank sort of lss.py:339-343) and the offsets a binary search - static
        shapes, no host synchronisation.  Entry id = pixel*D + d, pixel = ((b*N + n)*fH + h)*fW + w."""
        if has_transformation(img_metas):       # lss.py:262-265: the augmentation flow is affine - folded into the camera poses
            rots, trans = fold_into_cam2ego(rots, trans, img_metas)
        B, N = trans.shape[:2]
        inv, pt = self._aug(img_metas, post_rots, post_trans, rots)
        fr = self.frustum
        lower = (self.bx - self.dx / 2.0).tolist()
        X, Y, Z = (int(v) for v in self.nx)
        keys = ops.lss_cells(rots.contiguous(), trans.contiguous(), fr[0, 0, :, 0].contiguous(), fr[0, :, 0, 1].contiguous(),
                             fr[:, 0, 0, 2].contiguous(), lower, self.dx.tolist(), (X, Y, Z), inv, pt,
                             None if extra_rots is None else extra_rots.contiguous(),
                             None if extra_trans is None else extra_trans.contiguous())
        n_cells = B * Z * X * Y
        skeys, order = torch.sort(keys, stable=True)
        bounds = torch.arange(n_cells + 1, device=keys.device, dtype=torch.int32)
        offsets = torch.searchsorted(skeys, bounds, out_int32=True)
        return order.int(), offsets, n_cells

    def get_voxels(self, x, rots=None, trans=None, post_rots=None, post_trans=None, extra_rots=None, extra_trans=None,
                   img_metas=None):
        """-> (voxels (B, camC, Z, X, Y), depth (B, N, D, fH, fW)); lss.py:364-369."""
        B, N, Cin, H, W = x.shape
        x_cl = ops.nchw_to_nhwc(x.contiguous().view(B * N, Cin, H, W)).view(B * N * H * W, Cin)
        dn, D, Cc = self.camencode.depthnet, self.D, self.camC
        # one GEMM per pixel row: [features | depth logits | pad to a multiple of 4] so feature rows are 16-byte aligned
        pad = (-(D + Cc)) % 4
        w = dn.weight.view(D + Cc, Cin)
        w = torch.cat((w[D:], w[:D], w.new_zeros(pad, Cin)), 0)
        b = torch.cat((dn.bias[D:], dn.bias[:D], dn.bias.new_zeros(pad)), 0)
        ops.note_vendor('LSS depthnet', x_cl.shape[0], w.shape[0], Cin)
        y = F.linear(x_cl, w, b)                                                          # (P, camC + D + pad)
        depth = torch.softmax(y[:, Cc:Cc + D], dim=1).contiguous()                        # lss.py:132-133
        src, offsets, n_cells = self.cell_table(rots, trans, post_rots, post_trans, extra_rots, extra_trans, img_metas)
        X, Y, Z = (int(v) for v in self.nx)
        vox = ops.lss_splat(y[:, :Cc], depth, src, offsets, n_cells)
        vox = vox.view(B, Z, X, Y, self.camC).permute(0, 4, 1, 2, 3)
        return vox, depth.view(B, N, H, W, self.D).permute(0, 1, 4, 2, 3)

    def s2c(self, x):
        """lss.py:371-375."""
        B, C, H, W, L = x.shape
        return torch.reshape(x, (B, C * H, W, L)).permute((0, 1, 3, 2))

    def forward(self, x, rots, trans, lidar2img_rt=None, img_metas=None, post_rots=None, post_trans=None,
                extra_rots=None, extra_trans=None):
        if not x.is_cuda:
            raise RuntimeError('LiftSplatShoot: inputs must live on the MI355X (HIP) device - no CPU fallback')
        if self.training:
            return self._forward_train(x, rots, trans, img_metas, post_rots, post_trans, extra_rots, extra_trans)
        with torch.no_grad():
            vox, depth = self.get_voxels(x, rots, trans, post_rots, post_trans, extra_rots, extra_trans, img_metas)
            bev = self.s2c(vox).contiguous()
            folded = self._folded_bevencode()
            if self.dense_mode == 'f16x3' and all(w.shape[1] % 32 == 0 for w, _ in folded):
                # the four 3x3 convs on the split-fp16 MFMA kernels, chained through (hi, lo') NHWC pairs (convhalo.hip)
                if getattr(self, '_hints', None) is None or self._hints[0].device != bev.device:
                    self._hints = [ops.new_hint(bev.device) for _ in range(len(folded))]     # persistent exponent guesses
                pair = ops.split_f16(bev, to_nhwc=True, hint=self._hints[0])
                for i, (w, shift) in enumerate(folded):
                    last = i + 1 == len(folded)
                    pair = ops.conv3x3_f16x3(pair, self._split_w[i], shift, True, 1, split_out=not last and w.shape[0] % 2 == 0)
                    if not last and torch.is_tensor(pair):
                        pair = ops.split_f16(pair, to_nhwc=True, hint=self._hints[i + 1])
                return pair, depth
            for w, shift in folded:
                ops.note_vendor('LSS bevencode conv3x3', bev.shape[0] * bev.shape[2] * bev.shape[3], w.shape[0], 9 * w.shape[1])
                bev = ops.bias_relu_(F.conv2d(bev, w, None, padding=1), shift)
            return bev, depth

    def _forward_train(self, x, rots, trans, img_metas=None, post_rots=None, post_trans=None, extra_rots=None, extra_trans=None):
        """lss.py:377-383 under autograd (training mode: BatchNorm on batch statistics).  The depth net, the depth-weighted outer
        product (lss.py:135-141) and the BEV encoder are the framework's ops on the module's own parameters; the voxel pooling
        (lss.py:285-362) is ``autograd.bev_pool`` = ``ff3d_bev_pool`` / ``ff3d_bev_pool_bwd``, the MI355X counterparts of the
        reference's bev_pool extension - the same per-cell sums its default cumsum trick produces, for either ``newbevpool``."""
        from .autograd import bev_pool
        B, N, Cin, H, W = x.shape
        D, Cc = self.D, self.camC
        y = self.camencode.depthnet(x.view(B * N, Cin, H, W))
        depth = y[:, :D].softmax(dim=1)                                                    # lss.py:132-133
        feat = depth.unsqueeze(1) * y[:, D:D + Cc].unsqueeze(2)                            # (BN, camC, D, H, W)
        feat = feat.view(B, N, Cc, D, H, W).permute(0, 1, 3, 4, 5, 2).reshape(-1, Cc)      # lss.py:277-283
        with torch.no_grad():
            geom = self.get_geometry(rots, trans, post_rots, post_trans, extra_rots, extra_trans, img_metas)
            dx, bx, nx = self.dx.to(geom.device), self.bx.to(geom.device), self.nx.to(geom.device)
            cell = ((geom - (bx - dx / 2.0)) / dx).long().view(-1, 3)                      # lss.py:330
            batch_ix = torch.arange(B, device=geom.device).repeat_interleave(N * D * H * W)
            kept = ((cell[:, 0] >= 0) & (cell[:, 0] < nx[0]) & (cell[:, 1] >= 0) & (cell[:, 1] < nx[1])
                    & (cell[:, 2] >= 0) & (cell[:, 2] < nx[2]))
            coords = torch.cat((cell, batch_ix[:, None]), 1)[kept]
        X, Y, Z = (int(v) for v in self.nx)
        vox = bev_pool(feat[kept], coords, B, Z, X, Y)                                     # (B, camC, Z, X, Y)
        bev = self.bevencode(self.s2c(vox))
        return bev, depth.view(B, N, D, H, W)

    def _folded_bevencode(self):
        """BatchNorm folded into the four BEV-encoder convs, cached until a parameter / buffer changes."""
        mods = list(self.bevencode)
        sig = tuple((t.data_ptr(), t._version) for m in mods for t in list(m.parameters()) + list(m.buffers()))
        if getattr(self, '_fold_sig', None) != sig:
            folded = []
            for i in range(0, len(mods), 3):
                conv, bn = mods[i], mods[i + 1]
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                folded.append(((conv.weight * scale.view(-1, 1, 1, 1)).contiguous(),
                               (bn.bias - bn.running_mean * scale).contiguous()))
            self._fold_sig, self._folded = sig, folded
            self._split_w = [ops.split_weight_f16(w, bias=b) for w, b in folded] if folded[0][0].is_cuda else None
        return self._folded
