"""Import shims that let the *unmodified* reference head run on CPU in the build container.

TEST INFRASTRUCTURE ONLY (golden-vector generation).  Used by ``oracle/gen_golden.py``
in this container, where /root/reference exists; never on the GPU box, never by the
product.  Nothing from the reference is copied: the reference modules are imported from
where they lie, and the third-party packages they need but which are absent here
(mmcv-full 1.3.18, mmdet 2.14.0, mmdet3d 0.17.1 - doc/install.md:9-14) are replaced by the
minimal stand-ins below, restated from SURVEY.md Appendix A.
"""
import contextlib
import importlib
import importlib.machinery
import os
import sys
import types

import torch
import torch.nn as nn

sys.dont_write_bytecode = True   # never drop __pycache__ into /root/reference

REF_ROOT = '/root/reference'


class Registry:
    def __init__(self, name):
        self.name, self.d = name, {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.d[name or cls.__name__] = cls
            return cls
        return deco

    def build(self, cfg, **default):
        cfg = dict(cfg)
        cfg.update({k: v for k, v in default.items() if k not in cfg})
        return self.d[cfg.pop('type')](**cfg)


HEADS, BBOX_CODERS, TRANSFORMER, NECKS = Registry('head'), Registry('coder'), Registry('transformer'), Registry('neck')
BBOX_ASSIGNERS, MATCH_COST = Registry('assigner'), Registry('match cost')


class ConvModule(nn.Module):
    """A.4: conv -> norm -> act; bias='auto' => bias = not with_norm."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, bias='auto',
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), **kw):
        super().__init__()
        ctype = (conv_cfg or dict(type='Conv2d'))['type']
        with_norm = norm_cfg is not None
        if bias == 'auto':
            bias = not with_norm
        conv = {'Conv1d': nn.Conv1d, 'Conv2d': nn.Conv2d}[ctype]
        self.conv = conv(cin, cout, kernel_size, stride=stride, padding=padding, bias=bias)
        if with_norm:
            self.bn = {'BN1d': nn.BatchNorm1d, 'BN2d': nn.BatchNorm2d, 'BN': nn.BatchNorm2d}[norm_cfg['type']](cout)
        self.with_norm = with_norm
        self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        # The reference adds the RoI feature IN PLACE to the permuted decoder output (FD:921) that the previous stage's prediction
        # heads convolved.  The torch the reference ran on differentiated the convolution at the backend level, on the contiguous
        # copy it made of that view; today's torch keeps the view itself and refuses the backward.  The copy is restored here.
        x = self.conv(x if x.is_contiguous() or not torch.is_grad_enabled() else x.contiguous())
        if self.with_norm:
            x = self.bn(x)
        return self.activate(x)


def build_conv_layer(cfg, *a, **k):
    ctype = (cfg or dict(type='Conv2d'))['type']
    return {'Conv1d': nn.Conv1d, 'Conv2d': nn.Conv2d}[ctype](*a, **k)


class _MSDAParams(nn.Module):
    def __init__(self, C, heads, L, P):
        super().__init__()
        self.sampling_offsets = nn.Linear(C, heads * L * P * 2)
        self.attention_weights = nn.Linear(C, heads * L * P)
        self.value_proj = nn.Linear(C, C)
        self.output_proj = nn.Linear(C, C)


class _MHAParams(nn.Module):
    def __init__(self, C, heads):
        super().__init__()
        self.attn = nn.MultiheadAttention(C, heads, 0.1)


class _FFNParams(nn.Module):
    def __init__(self, C, F_):
        super().__init__()
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(C, F_), nn.ReLU(inplace=True), nn.Dropout(0.1)),
                                    nn.Linear(F_, C), nn.Dropout(0.1))


class _LayerParams(nn.Module):
    def __init__(self, C, heads, L, P, F_):
        super().__init__()
        self.attentions = nn.ModuleList([_MHAParams(C, heads), _MSDAParams(C, heads, L, P)])
        self.ffns = nn.ModuleList([_FFNParams(C, F_)])
        self.norms = nn.ModuleList([nn.LayerNorm(C) for _ in range(3)])


class ShimDeformableDecoder(nn.Module):
    """Parameter container with the mmcv/mmdet key layout whose forward is
    the oracle's restatement of A.1-A.3 - the reference has no source for this module."""

    def __init__(self, cfg):
        super().__init__()
        tl = cfg['transformerlayers']
        a0, a1 = tl['attn_cfgs']
        self.C, self.heads = a0['embed_dims'], a0['num_heads']
        self.L, self.P = a1['num_levels'], a1['num_points']
        self.num_layers = cfg['num_layers']
        assert tuple(tl['operation_order']) == ('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')
        self.layers = nn.ModuleList([_LayerParams(self.C, self.heads, self.L, self.P, tl['feedforward_channels'])
                                     for _ in range(self.num_layers)])
        self.taps = None

    def forward(self, query, key=None, value=None, query_pos=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, valid_ratios=None, key_padding_mask=None, attn_masks=None, **kw):
        from oracle import ff3d_oracle as O
        cfg = O.head_config(num_heads=self.heads, num_levels=self.L, num_points=self.P, num_layers=self.num_layers)
        shapes = [tuple(int(v) for v in s) for s in spatial_shapes.tolist()]
        taps = [] if self.taps is not None else None
        out = O.deformable_decoder(query, value, query_pos, reference_points, shapes, valid_ratios,
                                   dict(self.state_dict(keep_vars=True)), '', cfg, attn_mask=attn_masks, taps=taps)
        if taps is not None:
            self.taps.append(taps)
        return out


def rotation_3d_in_axis(points, angles, axis=0):
    """A.5 (mmdet3d v0.17.1)."""
    assert axis in (2, -1)
    s, c = torch.sin(angles), torch.cos(angles)
    o, z = torch.ones_like(c), torch.zeros_like(c)
    rot_mat_T = torch.stack([torch.stack([c, -s, z]), torch.stack([s, c, z]), torch.stack([z, z, o])])
    return torch.einsum('aij,jka->aik', (points, rot_mat_T))


class LiDARInstance3DBoxes:
    """Stand-in for mmdet3d 0.17.1 ``LiDARInstance3DBoxes`` (un-vendored): the container surface the reference's glue code
    touches - ``.tensor``, ``.bev`` (x, y, x_size, y_size, yaw), ``cat``, indexing, ``clone``, ``flip`` ('horizontal' negates
    y / vy and maps yaw -> -yaw + pi, 'vertical' negates x / vx and maps yaw -> -yaw), ``scale`` (all metric columns), ``to``."""

    def __init__(self, tensor, box_dim=7, **kw):
        self.tensor, self.box_dim = tensor, box_dim

    @property
    def bev(self):
        return self.tensor[:, [0, 1, 3, 4, 6]]

    @property
    def gravity_center(self):
        """mmdet3d: bottom centre + half the height."""
        t = self.tensor
        return torch.cat([t[:, :2], (t[:, 2] + t[:, 5] * 0.5)[:, None]], 1)

    @property
    def corners(self):
        """mmdet3d 0.17.1 ``LiDARInstance3DBoxes.corners`` (N, 8, 3): unit-cube corners in the order
        unravel_index(arange(8), (2,2,2))[[0,1,3,2,4,5,7,6]] minus (0.5, 0.5, 0), times (x_size, y_size, z_size), rotated by
        yaw about z (rotation_3d_in_axis), translated to the bottom centre."""
        import numpy as np
        t = self.tensor
        unit = torch.from_numpy(np.stack(np.unravel_index(np.arange(8), [2] * 3), axis=1)).to(t.dtype)
        unit = unit[[0, 1, 3, 2, 4, 5, 7, 6]] - t.new_tensor([0.5, 0.5, 0])
        c = t[:, 3:6].view(-1, 1, 3) * unit.reshape(1, 8, 3)
        c = rotation_3d_in_axis(c, t[:, 6], axis=2)
        return c + t[:, :3].view(-1, 1, 3)

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        t = self.tensor[item]
        return LiDARInstance3DBoxes(t.view(1, -1) if t.dim() == 1 else t, box_dim=self.box_dim)

    @classmethod
    def cat(cls, boxes_list):
        return cls(torch.cat([b.tensor for b in boxes_list], 0), box_dim=boxes_list[0].box_dim)

    def clone(self):
        return LiDARInstance3DBoxes(self.tensor.clone(), box_dim=self.box_dim)

    def to(self, device):
        return LiDARInstance3DBoxes(self.tensor.to(device), box_dim=self.box_dim)

    def flip(self, bev_direction='horizontal'):
        import math
        if bev_direction == 'horizontal':
            self.tensor[:, 1::7] = -self.tensor[:, 1::7]
            self.tensor[:, 6] = -self.tensor[:, 6] + math.pi
        else:
            self.tensor[:, 0::7] = -self.tensor[:, 0::7]
            self.tensor[:, 6] = -self.tensor[:, 6]

    def scale(self, scale_factor):
        self.tensor[:, :6] *= scale_factor
        self.tensor[:, 7:] *= scale_factor


def shim_bbox3d_mapping_back(bboxes, scale_factor, flip_horizontal, flip_vertical):
    """mmdet3d 0.17.1 ``bbox3d_mapping_back`` (un-vendored): undo the flips, then the scale, on a clone."""
    new_bboxes = bboxes.clone()
    if flip_horizontal:
        new_bboxes.flip('horizontal')
    if flip_vertical:
        new_bboxes.flip('vertical')
    new_bboxes.scale(1 / scale_factor)
    return new_bboxes


def shim_xywhr2xyxyr(boxes_xywhr):
    """mmdet3d ``xywhr2xyxyr`` (un-vendored)."""
    boxes = torch.zeros_like(boxes_xywhr)
    half_w, half_h = boxes_xywhr[:, 2] / 2, boxes_xywhr[:, 3] / 2
    boxes[:, 0], boxes[:, 1] = boxes_xywhr[:, 0] - half_w, boxes_xywhr[:, 1] - half_h
    boxes[:, 2], boxes[:, 3] = boxes_xywhr[:, 0] + half_w, boxes_xywhr[:, 1] + half_h
    boxes[:, 4] = boxes_xywhr[:, 4]
    return boxes


def shim_bbox3d2result(bboxes, scores, labels):
    """mmdet3d ``bbox3d2result`` (un-vendored): the result dict on the host."""
    return dict(boxes_3d=bboxes.to('cpu'), scores_3d=scores.cpu(), labels_3d=labels.cpu())


class AttrDict(dict):
    """mmcv ``Config``-like test_cfg: item and attribute access, ``copy`` keeps the type."""
    __setattr__ = dict.__setitem__

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def copy(self):
        return AttrDict(self)


class ShimInvertedResidual(nn.Module):
    """Parameter container with torchvision's ``mobilenetv2.InvertedResidual`` key layout; forward = the oracle's
    restatement (torchvision is not installed here)."""

    def __init__(self, inp, oup, stride, expand_ratio, norm_layer=None):
        super().__init__()
        assert stride == 1
        self.inp, self.oup, self.expand_ratio = inp, oup, expand_ratio
        hidden = int(round(inp * expand_ratio))
        cbr = lambda i, o, k, g: nn.Sequential(nn.Conv2d(i, o, k, 1, (k - 1) // 2, groups=g, bias=False),
                                              nn.BatchNorm2d(o), nn.ReLU6(inplace=True))
        layers = [cbr(inp, hidden, 1, 1)] if expand_ratio != 1 else []
        layers += [cbr(hidden, hidden, 3, hidden), nn.Conv2d(hidden, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        from oracle import ff3d_oracle as O
        return O.inverted_residual(x, dict(self.state_dict()), '', self.inp, self.oup, self.expand_ratio)


class ShimBasicBlock(nn.Module):
    """torchvision ``resnet.BasicBlock`` key layout; forward = the oracle's restatement."""

    def __init__(self, inplanes, planes, norm_layer=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)

    def forward(self, x):
        from oracle import ff3d_oracle as O
        return O.basic_block(x, dict(self.state_dict()), '')


def _mod(name, **attrs):
    m = types.ModuleType(name)
    # a real spec: libraries that probe optional dependencies with importlib.util.find_spec (HF transformers does so for
    # torchvision) raise "ValueError: <module>.__spec__ is None" on a bare ModuleType
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None, is_package=True)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def _na(*a, **k):
    raise NotImplementedError('third-party op not available in the shim harness')


def install():
    """Register the stand-in modules; idempotent."""
    if 'mmcv' in sys.modules and getattr(sys.modules['mmcv'], '_ff3d_shim', False):
        return
    ident = lambda *a, **k: (lambda f: f)
    _mod('mmcv', _ff3d_shim=True)
    _mod('mmcv.cnn', ConvModule=ConvModule, build_conv_layer=build_conv_layer, kaiming_init=_na, Linear=nn.Linear,
         build_activation_layer=_na, build_norm_layer=_na, xavier_init=_na)
    _mod('mmcv.cnn.bricks')
    _mod('mmcv.cnn.bricks.transformer', build_transformer_layer_sequence=lambda cfg: ShimDeformableDecoder(cfg))
    _mod('mmcv.runner', force_fp32=ident)
    _mod('mmdet')
    # training-side third party (mmdet 2.14 / mmdet3d 0.17.1, un-vendored): served by the restatements of oracle/train_oracle.py
    from oracle import train_oracle as _T
    MATCH_COST.d.setdefault('FocalLossCost', _T.FocalLossCost)
    _mod('mmdet.core', build_bbox_coder=lambda cfg: BBOX_CODERS.build(cfg), multi_apply=_T.multi_apply,
         build_assigner=lambda cfg: BBOX_ASSIGNERS.build(cfg), build_sampler=lambda cfg, **kw: _T.PseudoSampler(),
         AssignResult=_T.AssignResult)
    _mod('mmdet.core.bbox', BaseBBoxCoder=object)
    _mod('mmdet.core.bbox.builder', BBOX_CODERS=BBOX_CODERS, BBOX_ASSIGNERS=BBOX_ASSIGNERS)
    _mod('mmdet.core.bbox.assigners', AssignResult=_T.AssignResult, BaseAssigner=object)
    _mod('mmdet.core.bbox.match_costs', build_match_cost=lambda cfg: MATCH_COST.build(cfg))
    _mod('mmdet.core.bbox.match_costs.builder', MATCH_COST=MATCH_COST)
    _mod('mmdet.core.bbox.iou_calculators', build_iou_calculator=lambda cfg: _T.BboxOverlaps3D(**{k: v for k, v in cfg.items() if k != 'type'}))
    _mod('mmdet.models')
    _mod('mmdet.models.utils')
    _mod('mmdet.models.utils.builder', TRANSFORMER=TRANSFORMER)
    builder = _mod('mmdet3d.models.builder', HEADS=HEADS, NECKS=NECKS, build_loss=_T.build_loss, build_head=_na)
    _mod('mmdet3d')
    _mod('mmdet3d.models', builder=builder)
    _mod('mmdet3d.models.utils', clip_sigmoid=_T.clip_sigmoid)
    _mod('mmdet3d.models.fusion_layers', apply_3d_transformation=lambda pts, coord, meta, reverse=False: pts)
    core = _mod('mmdet3d.core', circle_nms=_na, draw_heatmap_gaussian=_T.draw_heatmap_gaussian, gaussian_radius=_T.gaussian_radius,
         xywhr2xyxyr=shim_xywhr2xyxyr, PseudoSampler=_T.PseudoSampler, LiDARInstance3DBoxes=LiDARInstance3DBoxes)
    _mod('mmdet3d.core.bbox', bbox3d2result=shim_bbox3d2result, bbox3d_mapping_back=shim_bbox3d_mapping_back,
         xywhr2xyxyr=shim_xywhr2xyxyr, CameraInstance3DBoxes=object, DepthInstance3DBoxes=object,
         LiDARInstance3DBoxes=LiDARInstance3DBoxes, box_np_ops=None)
    _mod('mmdet3d.core.bbox.structures')
    _mod('mmdet3d.core.bbox.structures.utils', rotation_3d_in_axis=rotation_3d_in_axis)
    _mod('mmdet3d.ops')
    _mod('mmdet3d.ops.iou3d')
    from oracle import ff3d_oracle as _O

    def _circle_nms(dets, thresh, post_max_size=83):
        # mmdet3d's numba `circle_nms` (un-vendored) served by the oracle's restatement of its published algorithm
        return _O.circle_nms(dets, thresh, post_max_size)
    core.circle_nms = _circle_nms

    def _nms_gpu(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
        # mmdet3d's CUDA `nms_gpu` (un-vendored) served by the oracle's restatement of iou3d_kernel.cu
        keep = _O.nms_bev(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), thresh, pre_maxsize, post_max_size)
        return torch.tensor(keep, dtype=torch.long, device=boxes.device)

    def _boxes_iou_bev(a, b):
        return torch.from_numpy(_O.boxes_iou_bev(a.detach().cpu().numpy(), b.detach().cpu().numpy())).to(a.device)
    _mod('mmdet3d.ops.iou3d.iou3d_utils', nms_gpu=_nms_gpu, nms_normal_gpu=_na, boxes_iou_bev=_boxes_iou_bev)
    # mmdet3d's CUDA `points_in_boxes_gpu` (un-vendored; FD:742, mask_heatmap_mode='boxcls') served by the oracle's restatement of
    # points_in_boxes_cuda.cu
    _mod('mmdet3d.ops.roiaware_pool3d', points_in_boxes_gpu=lambda points, boxes: _O.points_in_boxes(points, boxes))
    base = REF_ROOT + '/projects'
    _pkg('projects', base)
    _pkg('projects.mmdet3d_plugin', base + '/mmdet3d_plugin')
    _pkg('projects.mmdet3d_plugin.models', base + '/mmdet3d_plugin/models')
    _pkg('projects.mmdet3d_plugin.models.dense_heads', base + '/mmdet3d_plugin/models/dense_heads')
    u = _pkg('projects.mmdet3d_plugin.models.utils', base + '/mmdet3d_plugin/models/utils')
    from oracle import ff3d_oracle as O
    # the reference's CUDA-only extension (ops/locatt_ops/__init__.py:11 asserts CUDA) -> the oracle's restatement of
    # kernels.cuh, so LocalContextAttentionBlock / FocalEncoder glue code of the reference can run on CPU
    locatt = types.SimpleNamespace(localattention=types.SimpleNamespace(similar_forward=O.locatt_similar,
                                                                        weighting_forward=O.locatt_weighting))
    ops = _mod('projects.mmdet3d_plugin.models.utils.ops', locatt_ops=locatt)
    u.ops = ops
    _mod('torchvision')
    _mod('torchvision.models')
    _mod('torchvision.models.resnet', BasicBlock=ShimBasicBlock)
    _mod('torchvision.models.mobilenetv2', InvertedResidual=ShimInvertedResidual)
    sys.modules['torchvision.models'].resnet = sys.modules['torchvision.models.resnet']
    sys.modules['torchvision.models'].mobilenetv2 = sys.modules['torchvision.models.mobilenetv2']
    sys.modules['torchvision'].models = sys.modules['torchvision.models']
    _pkg('projects.mmdet3d_plugin.models.necks', base + '/mmdet3d_plugin/models/necks')
    # import-time dependencies of necks/lss.py that are absent here and unused by the inference path
    sys.modules['torchvision.models.resnet'].resnet18 = _na
    _mod('torchvision.utils', save_image=_na)
    sys.modules['torchvision'].utils = sys.modules['torchvision.utils']
    _mod('matplotlib')
    _mod('matplotlib.pyplot')
    sys.modules['matplotlib'].pyplot = sys.modules['matplotlib.pyplot']
    _mod('mpl_toolkits')
    _mod('mpl_toolkits.mplot3d', Axes3D=object)
    _pkg('projects.mmdet3d_plugin.core', base + '/mmdet3d_plugin/core')
    _pkg('projects.mmdet3d_plugin.core.bbox', base + '/mmdet3d_plugin/core/bbox')
    _pkg('projects.mmdet3d_plugin.core.bbox.coders', base + '/mmdet3d_plugin/core/bbox/coders')
    _pkg('projects.mmdet3d_plugin.core.bbox.assigners', base + '/mmdet3d_plugin/core/bbox/assigners')
    _pkg('projects.mmdet3d_plugin.core.post_processing', base + '/mmdet3d_plugin/core/post_processing')
    sys.modules['mmcv'].mkdir_or_exist = lambda d: os.makedirs(d, exist_ok=True)


def load_reference():
    """Import the reference modules on the path, unmodified.  Returns a namespace."""
    install()
    fd = importlib.import_module('projects.mmdet3d_plugin.models.dense_heads.focal_decoder')
    bc = importlib.import_module('projects.mmdet3d_plugin.core.bbox.coders.transfusion_bbox_coder')
    eu = importlib.import_module('projects.mmdet3d_plugin.models.utils.encoder_utils')
    ut = importlib.import_module('projects.mmdet3d_plugin.models.utils.utils')
    fe = importlib.import_module('projects.mmdet3d_plugin.models.necks.focal_encoder')
    lss = importlib.import_module('projects.mmdet3d_plugin.models.necks.lss')
    ma = importlib.import_module('projects.mmdet3d_plugin.core.post_processing.merge_augs')
    ha = importlib.import_module('projects.mmdet3d_plugin.core.bbox.assigners.hungarian_assigner')   # registers HungarianAssigner3D + costs
    return types.SimpleNamespace(merge_augs=ma, hungarian_assigner=ha, FocalDecoder=fd.FocalDecoder, TransFusionBBoxCoder=bc.TransFusionBBoxCoder,
                                 I2P=eu.I2P, utils=ut, fd=fd, eu=eu, FocalEncoder=fe.FocalEncoder,
                                 LocalContextAttentionBlock=eu.LocalContextAttentionBlock,
                                 LiftSplatShoot=lss.LiftSplatShoot)


@contextlib.contextmanager
def cpu_device_patch(rand_log=None):
    """The reference hard-codes device='cuda' (FD:380,384,408,837-841,851,863,904) and calls .cuda() (FD:395,397; EU:172,182,
    204); map both to CPU while the reference code runs.  ``rand_log``: a list that receives every torch.rand draw (the
    ground-truth-group noise of the training-mode forward, FD:408) so that a test can replay it."""
    orig_as, orig_ones, orig_zeros, orig_rand, orig_cuda = torch.as_tensor, torch.ones, torch.zeros, torch.rand, torch.Tensor.cuda

    def fix(kw):
        if str(kw.get('device', '')).startswith('cuda'):
            kw['device'] = 'cpu'
        return kw

    def rand(*a, **k):
        r = orig_rand(*a, **fix(k))
        if rand_log is not None:
            rand_log.append(r.clone())
        return r
    torch.as_tensor = lambda *a, **k: orig_as(*a, **fix(k))
    torch.ones = lambda *a, **k: orig_ones(*a, **fix(k))
    torch.zeros = lambda *a, **k: orig_zeros(*a, **fix(k))
    torch.rand = rand
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.as_tensor, torch.ones, torch.zeros, torch.rand, torch.Tensor.cuda = orig_as, orig_ones, orig_zeros, orig_rand, orig_cuda
