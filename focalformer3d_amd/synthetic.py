"""Synthetic workloads of BASELINE.json (random-init weights of the reference architecture + N(0,1)
feature maps; there are no datasets or checkpoints in this environment).  Shared by bench.py, the
smoke test and the GPU tests so they all exercise the same configuration."""
import torch

from .registry import build_head
from . import focal_decoder as _fd  # noqa: F401


def decoder_cfg(C, num_layers=3, ffn=1024, num_levels=3, num_points=4, heads=8):
    """The ``decoder_cfg`` of FocalFormer3D_L.py:285-313 at hidden width C."""
    return dict(type='DeformableDetrTransformerDecoder', num_layers=num_layers, return_intermediate=False,
                transformerlayers=dict(
                    type='DetrTransformerDecoderLayer',
                    attn_cfgs=[dict(type='MultiheadAttention', embed_dims=C, num_heads=heads, dropout=0.1),
                               dict(type='MultiScaleDeformableAttention', embed_dims=C, num_levels=num_levels,
                                    num_points=num_points, num_heads=heads)],
                    feedforward_channels=ffn, ffn_dropout=0.1,
                    ffn_cfgs=dict(type='FFN', embed_dims=C, num_fcs=2, act_cfg=dict(type='ReLU', inplace=True)),
                    operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))


def focalformer3d_l_head_cfg(C=256, grid=180, num_proposals=200, stages=3, decoder_stages=2, num_classes=10,
                             dataset='nuScenes', ffn=1024, hidden_channel_roi=512):
    """``pts_bbox_head`` dict in the layout of FocalFormer3D_L.py:238-314 with BASELINE.json's shape:
    ``stages`` HIP stages (reuse_first_heatmap => multistage_heatmap = stages-1) x ``num_proposals`` queries."""
    nus = dataset == 'nuScenes'
    pcr = [-54.0, -54.0] if nus else [-75.2, -75.2]
    vox = 2 * abs(pcr[0]) / (grid * 8)
    heads = dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2))
    if nus:
        heads['vel'] = (2, 2)
    return dict(
        type='FocalDecoder', reuse_first_heatmap=True, extra_feat=True, roi_feats=7, roi_dropout_rate=0.1,
        roi_based_reg=True, roi_expand_ratio=1.2, heatmap_box=False, thin_heatmap_box=False, multiscale=True,
        multistage_heatmap=stages - 1, mask_heatmap_mode='poscls', input_img=False, iterbev_wo_img=True,
        add_gt_groups=3, bevpos=True, num_proposals=num_proposals, hidden_channel=C, hidden_channel_roi=hidden_channel_roi,
        num_classes=num_classes, num_decoder_layers=decoder_stages, num_heads=8, initialize_by_heatmap=True,
        nms_kernel_size=3, bn_momentum=0.1, activation='relu', common_heads=heads,
        bbox_coder=dict(type='TransFusionBBoxCoder', pc_range=pcr, voxel_size=[vox, vox], out_size_factor=8,
                        post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0] if nus else [-80, -80, -10.0, 80, 80, 10.0],
                        score_threshold=0.0, code_size=10 if nus else 8),
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2, alpha=0.25, reduction='mean', loss_weight=1.0),
        loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=0.25),
        loss_heatmap=dict(type='GaussianFocalLoss', reduction='mean', loss_weight=1.0),
        decoder_cfg=decoder_cfg(C, ffn=ffn),
        test_cfg=dict(dataset=dataset, grid_size=[grid * 8, grid * 8, 40], out_size_factor=8, pc_range=pcr,
                      voxel_size=[vox, vox], nms_type=None))


def deformformer3d_l_head_cfg(C=256, grid=180, num_proposals=200):
    """``pts_bbox_head`` of DeformFormer3D_L.py:236-302 (BASELINE.json configs[0], SURVEY.md §8d "C1"): the single-stage
    branch FD:539-586 (``multistage_heatmap=None``), 200 queries, one decoder stage of 3 layers, no RoI branch."""
    cfg = focalformer3d_l_head_cfg(C=C, grid=grid, num_proposals=num_proposals, stages=1, decoder_stages=1)
    cfg.update(reuse_first_heatmap=False, extra_feat=False, roi_feats=0, roi_dropout_rate=0.0, roi_based_reg=False,
               roi_expand_ratio=1.0, multistage_heatmap=None)
    return cfg


def waymo_shape_head_cfg(C=256, grid=468, num_proposals=250, stages=4, **kw):
    """BASELINE.json configs[4]: Waymo-shape 468x468xC BEV (levels 468 / 234 / 117, Nv = 287 469), 1000 queries = 4 HIP
    stages x 250, K = 3 with kernel-1 classes 1, 2, no velocity head (FocalFormer3D_Waymo_L.py:193-227)."""
    return focalformer3d_l_head_cfg(C=C, grid=grid, num_proposals=num_proposals, stages=stages, decoder_stages=2,
                                    num_classes=3, dataset='Waymo', **kw)


def focalformer3d_lc_cfgs(C=256, Ci=256, grid=180, num_proposals=200, stages=3, pts_channels=512, max_points_height=10,
                          **head_kw):
    """BASELINE.json configs[2] (FocalFormer3D_LC_Proj.py:186-260 at BASELINE's widths): the fusion neck ``FocalEncoder`` with
    ``iterbev='bevfusion'`` (LocalContextAttentionBlock + the I2P camera-projection sampler, ``iter_bev_cam``) producing the
    head's stage maps, and the head without ``reuse_first_heatmap`` (``stages`` HIP stages x ``num_proposals`` queries).
    -> (neck cfg, head cfg)."""
    neck = dict(type='FocalEncoder', num_layers=stages, in_channels_img=Ci, in_channels_pts=pts_channels, hidden_channel=C,
                bn_momentum=0.1, max_points_height=max_points_height, bias='auto', iterbev='bevfusion', iter_bev_cam=True,
                multistage_heatmap=stages, extra_feat=True, input_img=True, input_pts=True, iterbev_wo_img=False, cam_lss=False)
    head = focalformer3d_l_head_cfg(C=C, grid=grid, num_proposals=num_proposals, stages=stages + 1, **head_kw)
    head.update(reuse_first_heatmap=False, input_img=True, add_gt_groups=0)
    return neck, head


def lc_inputs(B, Ci=256, grid=180, pts_channels=512, cam_hw=(232, 400), ncam=6, seed=0, device=None):
    """Synthetic inputs of configs[2]: ``ncam`` camera feature maps (B*ncam, Ci, 232, 400) ~ N(0,1) (FPN level 0 of 928x1600
    images), the LiDAR BEV map (B, pts_channels, grid, grid) ~ N(0,1) and img_metas with the synthetic pinhole rig."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B * ncam, Ci, *cam_hw, generator=g)
    pts = torch.randn(B, pts_channels, grid, grid, generator=g)
    shape = (cam_hw[0] * 4, cam_hw[1] * 4)
    l2i = camera_rig(B, ncam, shape)
    metas = [dict(lidar2img=l2i[b], input_shape=shape) for b in range(B)]
    if device is not None:
        img, pts = img.to(device), pts.to(device)
    return img, pts, metas, l2i


def build_neck_from_cfg(cfg, seed=0, device=None):
    from .focal_encoder import NECKS
    torch.manual_seed(seed)
    neck = randomize_(NECKS.build(dict(cfg)), seed).eval()
    return neck if device is None else neck.to(device)


def randomize_(module, seed=0):
    """Random weights of the architecture: module default init, decoder matrices xavier
    (FD:346-350, done by the head), BatchNorm running statistics randomised so BN is not the identity,
    heatmap biases at the reference's -2.19 so scores look like a trained head's."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, b in module.named_buffers():
            if n.endswith('running_mean'):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif n.endswith('running_var'):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
        for n, p in module.named_parameters():
            if n.endswith('sampling_offsets.weight') or n.endswith('attention_weights.weight'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)   # mmcv's zero init would make every query sample the same ring
    if hasattr(module, 'invalidate_cache'):
        module.invalidate_cache()
    return module


def build_head_from_cfg(cfg, seed=0, device=None):
    torch.manual_seed(seed)
    head = randomize_(build_head(cfg), seed).eval()
    return head if device is None else head.to(device)


def stage_features(B, C, grid, n_maps, seed=0, device=None):
    """[pts_feat_conv, [stage maps..., extra map]] ~ N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    f = [torch.randn(B, C, grid, grid, generator=g) for _ in range(n_maps + 1)]
    if device is not None:
        f = [t.to(device) for t in f]
    return [f[0], f[1:]]


def camera_rig(B, ncam, input_shape, height=1.0, radius=0.5, focal=0.79):
    """Synthetic ``lidar2img`` (B, ncam, 4, 4): ``ncam`` pinhole cameras looking outward at equal yaw spacing,
    focal length ``focal`` (0.79) * image width, principal point at the image centre."""
    import numpy as np
    Himg, Wimg = input_shape
    out = np.zeros((B, ncam, 4, 4), dtype=np.float32)
    for b in range(B):
        for c in range(ncam):
            yaw = 2 * np.pi * c / ncam + 0.05 * b
            fwd = np.array([np.cos(yaw), np.sin(yaw), 0.0])
            right = np.array([np.sin(yaw), -np.cos(yaw), 0.0])
            down = np.array([0.0, 0.0, -1.0])
            R = np.stack([right, down, fwd])
            t = -R @ np.array([radius * np.cos(yaw), radius * np.sin(yaw), height])
            f = focal * Wimg
            K = np.array([[f, 0, Wimg / 2], [0, f, Himg / 2], [0, 0, 1.0]])
            M = np.eye(4)
            M[:3, :3] = K @ R
            M[:3, 3] = K @ t
            out[b, c] = M
    return out
