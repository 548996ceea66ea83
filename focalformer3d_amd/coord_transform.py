"""Point-cloud augmentation bookkeeping of the camera branches: mmdet3d v0.17.1 ``apply_3d_transformation``
(mmdet3d/models/fusion_layers/coord_transform.py; un-vendored third party), as the reference calls it
from I2P (encoder_utils.py:222, ``reverse=True``: BEV pillar points back into the un-augmented LiDAR frame before they are
projected into the cameras) and from LiftSplatShoot (lss.py:262-265, ``reverse=False``: frustum points into the augmented
frame) - training-time augmentation and the flip / scale passes of test-time augmentation (focalformer3d.py:353-374).

The flow recorded in ``img_meta['transformation_3d_flow']`` is a sequence of 'R' (points @ pcd_rotation), 'S' (* pcd_scale_factor),
'T' (+ pcd_trans), 'HF' (y -> -y) and 'VF' (x -> -x) steps; reversed, the steps run backwards with their inverses.  Every step is
affine, so the whole flow is ONE affine map p -> p @ A + t.  MI355X design: that map is composed once per frame on the host (in
float64, 12 numbers) and folded into the matrices the kernels already take - ``lidar2img`` for the camera sampler, ``rots`` /
``trans`` for Lift-Splat-Shoot - instead of moving the (H*W*Z, 3) / (N*D*fH*fW, 3) point sets through five tensor ops.
"""
import numpy as np
import torch


def transformation_affine(img_meta, reverse=False):
    """-> (A (3, 3), t (3,)) float64 with apply_3d_transformation(p, 'LIDAR', img_meta, reverse) == p @ A + t."""
    rot = np.asarray(torch.as_tensor(img_meta['pcd_rotation']).cpu().numpy() if 'pcd_rotation' in img_meta else np.eye(3),
                     dtype=np.float64)
    scale = float(img_meta.get('pcd_scale_factor', 1.0))
    trans = np.asarray(torch.as_tensor(img_meta['pcd_trans']).cpu().numpy() if 'pcd_trans' in img_meta else np.zeros(3),
                       dtype=np.float64).reshape(3)
    hflip, vflip = bool(img_meta.get('pcd_horizontal_flip', False)), bool(img_meta.get('pcd_vertical_flip', False))
    flow = list(img_meta.get('transformation_3d_flow', []))
    if reverse:
        rot, scale, trans, flow = np.linalg.inv(rot), 1.0 / scale, -trans, flow[::-1]
    A, t = np.eye(3), np.zeros(3)
    for op in flow:
        if op == 'R':                                 # BasePoints.rotate with a matrix: points @ rotation
            A, t = A @ rot, t @ rot
        elif op == 'S':
            A, t = A * scale, t * scale
        elif op == 'T':
            t = t + trans
        elif op == 'HF':                              # LiDARPoints.flip('horizontal'): y -> -y (only if the meta says it happened)
            if hflip:
                A, t = A * np.array([1.0, -1.0, 1.0]), t * np.array([1.0, -1.0, 1.0])
        elif op == 'VF':
            if vflip:
                A, t = A * np.array([-1.0, 1.0, 1.0]), t * np.array([-1.0, 1.0, 1.0])
        else:
            raise AssertionError(f'This 3D data transformation op ({op}) is not supported')
    return A, t


def apply_3d_transformation(pcd, coord_type, img_meta, reverse=False):
    """mmdet3d's entry point (same signature): pcd (n, 3) LiDAR points -> transformed (n, 3) (a new tensor)."""
    if coord_type != 'LIDAR':
        raise NotImplementedError("FocalFormer3D only transforms 'LIDAR' points")
    A, t = transformation_affine(img_meta, reverse)
    return pcd @ torch.as_tensor(A, dtype=pcd.dtype, device=pcd.device) + torch.as_tensor(t, dtype=pcd.dtype, device=pcd.device)


def has_transformation(img_metas):
    return img_metas is not None and any(m.get('transformation_3d_flow') for m in img_metas)


def fold_into_lidar2img(lidar2img, img_meta):
    """lidar2img (Ncam, 4, 4) -> the matrices that project AUGMENTED-frame points: M' = M @ [[A^T, t], [0, 1]] with (A, t) the
    reverse flow (EU:222: the pillar grid lives in the augmented frame, the cameras in the original one)."""
    A, t = transformation_affine(img_meta, reverse=True)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = A.T, t
    return (np.asarray(lidar2img, dtype=np.float64) @ T).astype(np.float32)


def fold_into_cam2ego(rots, trans, img_metas):
    """LSS (lss.py:262-265): ego-frame frustum points R p + t are moved into the augmented frame, (R p + t) @ A + ta (row-vector
    form) = (A^T R) p + (A^T t + ta).  rots (B, N, 3, 3), trans (B, N, 3) -> the folded pair (same device / dtype)."""
    R, tt = rots.detach().cpu().double().numpy().copy(), trans.detach().cpu().double().numpy().copy()
    for b, meta in enumerate(img_metas):
        A, ta = transformation_affine(meta, reverse=False)
        R[b] = A.T @ R[b]
        tt[b] = tt[b] @ A + ta
    return torch.as_tensor(R, dtype=rots.dtype, device=rots.device), torch.as_tensor(tt, dtype=trans.dtype, device=trans.device)
