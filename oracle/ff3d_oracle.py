"""CPU restatement (PyTorch fp32 / numpy) of the FocalFormer3D HIP-decoder hot path.

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.  This file is the checker the
HIP path is compared against; it is never imported by ``focalformer3d_amd``.

Every function cites the reference lines it follows.  Abbreviations (paths relative
to /root/reference):

  FD = projects/mmdet3d_plugin/models/dense_heads/focal_decoder.py
  UT = projects/mmdet3d_plugin/models/utils/utils.py
  DU = projects/mmdet3d_plugin/models/utils/decoder_utils.py
  EU = projects/mmdet3d_plugin/models/utils/encoder_utils.py
  BC = projects/mmdet3d_plugin/core/bbox/coders/transfusion_bbox_coder.py
  A.x = SURVEY.md Appendix A: the published algorithm of the un-vendored third-party
        dependencies mmcv-full==1.3.18 / mmdet==2.14.0 / mmdet3d v0.17.1
        (pins: doc/install.md:9-14), restated because their source is not under
        /root/reference.

Pinning status (see DESIGN.md "Oracle"):
  * FD / UT / DU.FFN / BC / EU.I2P code paths: pinned against the reference itself,
    imported in the build container by ``oracle/gen_golden.py`` -> tests/golden/*.npz.
  * MSDA core (A.3): pinned against the independent HF ``transformers`` implementation
    of the same Deformable-DETR op.
  * Decoder layer / sequence / MSDA-module wiring (A.1 - A.3): no source and no reference test
    exists for it in /root/reference (mmcv / mmdet are un-vendored); restated from the
    published algorithm, anchored on the reference call site FD:927-933, and - round 6 -
    pinned by EXECUTION against the independent HF ``transformers`` implementation of the
    same Deformable-DETR decoder (``DeformableDetrDecoder`` / ``DeformableDetrDecoderLayer``
    holding the same mmcv-layout parameters: tests/golden/decoder_hf_{a,b,c}.npz, max
    deviation 1.7e-6; incl. per-level valid ratios and the bool self-attention mask of
    FD:851-856).  What stays a reading: mmcv's registry / config-dict plumbing only.

The weights are passed as a flat ``state_dict`` with the reference's own key names
 so one dict drives the reference module, this oracle and the
HIP product module.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
LN_EPS = 1e-5

# classes whose NMS / mask-dilation kernel is 1 instead of nms_kernel_size
# (FD:564-569, 678-683, 777-780)
SMALL_CLASSES = {'nuScenes': (8, 9), 'Waymo': (1, 2)}
# hard-coded RoI normalisation ranges (FD:903-906)
ROI_PC_RANGE = {'nuScenes': (-54.0, -54.0, 54.0, 54.0), 'Waymo': (-75.2, -75.2, 75.2, 75.2)}


def head_config(**kw):
    """Head hyper-parameters; defaults mirror FD:35-117 where the inference path reads them."""
    d = dict(
        num_proposals=128, hidden_channel=128, num_classes=4, num_decoder_layers=1,
        num_heads=8, nms_kernel_size=1, multiscale=False, multistage_heatmap=0,
        reuse_first_heatmap=False, extra_feat=False, bevpos=False, input_img=True,
        iterbev_wo_img=False, mask_heatmap_mode='poscls', roi_feats=0,
        roi_expand_ratio=1.0, roi_based_reg=False, classaware_reg=False,
        heatmap_box=False, thin_heatmap_box=False,               # FD:68-69 (only the thin form: the other needs DCNSeparateHead)
        common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
        dataset='nuScenes', num_levels=3, num_points=4, num_layers=3,
        # bbox coder (BC:10-22)
        pc_range=(-54.0, -54.0), voxel_size=(0.075, 0.075), out_size_factor=8,
        post_center_range=(-61.2, -61.2, -10.0, 61.2, 61.2, 10.0), score_threshold=0.0,
        # 'f32' (the reference's arithmetic) or 'bf16' = BASELINE.json configs[4] "bf16 QKV/FFN on MFMA": see lin()
        gemm_dtype='f32',
    )
    d.update(kw)
    cfg = SimpleNamespace(**d)
    # FD:138-139
    cfg.num_stages = int(cfg.multistage_heatmap or 0) + (1 if cfg.reuse_first_heatmap else 0)
    if isinstance(cfg.roi_expand_ratio, (int, float)):
        cfg.roi_expand_ratio = [float(cfg.roi_expand_ratio)] * cfg.num_decoder_layers  # FD:181-184
    return cfg


# --------------------------------------------------------------------------------------
# small building blocks
# --------------------------------------------------------------------------------------
def lin(x, w, b=None, lowp=False, relu=False):
    """``F.linear`` (+ ReLU).  ``lowp``: the arithmetic of BASELINE.json configs[4] ("bf16 QKV/FFN on MFMA") as the product's
    ``FocalDecoder.set_gemm_dtype(torch.bfloat16)`` defines it - input, weight and bias rounded to bf16 (round-to-nearest-even),
    products exact, fp32 accumulation, bias added in fp32, ONE rounding of the result to bf16 (a bf16-output MFMA GEMM), ReLU on
    the rounded value; returned as fp32.  The reference has no reduced-precision mode: this restates the GEMM-operand rounding
    only, everything else (softmax, LayerNorm, bilinear gathers, residuals) stays the reference's fp32 arithmetic."""
    if not lowp:
        y = F.linear(x, w, b)
        return F.relu(y) if relu else y
    r = lambda t: None if t is None else t.to(torch.bfloat16).float()
    y = F.linear(r(x), r(w), r(b)).to(torch.bfloat16).float()
    return F.relu(y) if relu else y


def conv_module_2d(x, sd, p, stride=1):
    """mmcv ConvModule(conv3x3 pad1 [no bias: bias='auto' with norm], BN2d eval, ReLU) - A.4;
    used at FD:151-162 (dconv/dconv2) and FD:204-212 (heatmap_head.0)."""
    y = F.conv2d(x, sd[p + 'conv.weight'], sd.get(p + 'conv.bias'), stride=stride, padding=1)
    y = F.batch_norm(y, sd[p + 'bn.running_mean'], sd[p + 'bn.running_var'],
                     sd[p + 'bn.weight'], sd[p + 'bn.bias'], False, 0.0, BN_EPS)
    return F.relu(y)


def heatmap_head(x, sd, p):
    """FD:202-221: ConvModule(C->C) then a bare biased Conv2d(C->K, 3x3, pad 1)
    (``bias='auto'`` is truthy for nn.Conv2d, A.4)."""
    y = conv_module_2d(x, sd, p + '0.')
    return F.conv2d(y, sd[p + '1.weight'], sd[p + '1.bias'], padding=1)


def create_2d_grid(x_size, y_size):
    """FD:337-344: cell centres, row-major over (y, x), columns (x+0.5, y+0.5)."""
    ys, xs = torch.meshgrid(torch.linspace(0, x_size - 1, x_size),
                            torch.linspace(0, y_size - 1, y_size), indexing='ij')
    return torch.stack([xs + 0.5, ys + 0.5], 0).view(1, 2, -1).permute(0, 2, 1)


def sine_dim_t():
    """UT:44-45."""
    dim_t = torch.arange(128, dtype=torch.float32)
    return 10000 ** (2 * (dim_t // 2) / 128)


def gen_sineembed_for_position(pos):
    """UT:40-53 (2-d branch): output order (y-embed, x-embed), 256-d."""
    scale = 2 * math.pi
    dim_t = sine_dim_t().to(pos.device)
    x_embed = pos[:, :, 0] * scale
    y_embed = pos[:, :, 1] * scale
    pos_x = x_embed[:, :, None] / dim_t
    pos_y = y_embed[:, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, 0::2].sin(), pos_x[:, :, 1::2].cos()), dim=3).flatten(2)
    pos_y = torch.stack((pos_y[:, :, 0::2].sin(), pos_y[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((pos_y, pos_x), dim=2)


def mlp(x, sd, p, num_layers=2):
    """UT:16-28."""
    for i in range(num_layers):
        x = F.linear(x, sd[f'{p}layers.{i}.weight'], sd[f'{p}layers.{i}.bias'])
        if i < num_layers - 1:
            x = F.relu(x)
    return x


# --------------------------------------------------------------------------------------
# Hard-instance-probing stage: sigmoid*mask -> NMS -> top-k -> gathers -> mask update
# --------------------------------------------------------------------------------------
def local_max_nms(heatmap, ksize, small_classes):
    """FD:672-685 (== FD:558-571).  ``heatmap`` (B,K,H,W) post-sigmoid post-mask."""
    pad = ksize // 2
    local_max = torch.zeros_like(heatmap)
    inner = F.max_pool2d(heatmap, kernel_size=ksize, stride=1, padding=0)
    if pad > 0:
        local_max[:, :, pad:-pad, pad:-pad] = inner
    else:
        local_max = inner
    for c in small_classes:
        if c < heatmap.shape[1]:
            local_max[:, c] = heatmap[:, c]
    return heatmap * (heatmap == local_max)


def topk_deterministic(flat, k):
    """FD:688 ``torch.topk(sorted=False)`` / FD:574 ``argsort(descending)[:k]``.

    The reference's choice among equal scores and its output order are implementation
    defined.  The oracle (and the HIP kernel) fix both: order = score descending, ties by
    lowest flat index.  Parity tests compare as sets and assert the k-th/(k+1)-th margin.
    """
    idx = torch.sort(flat, dim=-1, descending=True, stable=True).indices
    return idx[..., :k]


# heatmap_box branch: class -> task of the six CenterPoint-style task groups (FD:232-239: 1 + 2 + 2 + 1 + 2 + 2 classes)
HEATMAP_TASK_OF_CLASS = (0, 1, 1, 2, 2, 3, 4, 4, 5, 5)
# the clips of the dense heatmap boxes (FD:714-717); np.log in float64, compared in float32 like torch.clip does
HEATMAP_BOX_CLIPS = ((2, 3, -5.0, 3.0), (3, 6, float(np.log(0.5)), float(np.log(15))), (6, 8, -1.0, 1.0), (8, 10, -15.0, 15.0))


def points_in_boxes(points, boxes):
    """mmdet3d v0.17.1 ``points_in_boxes_gpu`` (mmdet3d/ops/roiaware_pool3d/src/points_in_boxes_cuda.cu, un-vendored: restated
    from the published kernel; call site FD:742,756-758).  points (B,M,3), boxes (B,T,7) = (cx, cy, cz bottom, w, l, h, rz) in
    LiDAR coordinates -> (B,M) int32: index of the FIRST box that contains the point, -1 for none.  The kernel shifts cz by
    h / 2, rejects |z - cz| > h / 2, rotates the offset by rz + pi / 2 and tests local_x in (-l/2, l/2), local_y in (-w/2, w/2)
    (strict).  float32 arithmetic as in the kernel."""
    p = points.detach().cpu().numpy().astype(np.float32)
    b = boxes.detach().cpu().numpy().astype(np.float32)
    B, M, _ = p.shape
    T = b.shape[1]
    out = np.full((B, M), -1, dtype=np.int32)
    two = np.float32(2.0)
    for bi in range(B):
        found = np.zeros(M, dtype=bool)
        for t in range(T):
            cx, cy, cz, w, l, h, rz = b[bi, t]
            czc = cz + h / two
            rot = np.float32(np.float64(rz) + np.pi / 2)        # `float rot_angle = rz + M_PI / 2` (sum in double, stored as float)
            cosa, sina = np.cos(rot, dtype=np.float32), np.sin(rot, dtype=np.float32)
            sx, sy = p[bi, :, 0] - cx, p[bi, :, 1] - cy
            lx = sx * cosa + sy * (-sina)
            ly = sx * sina + sy * cosa
            inside = (np.abs(p[bi, :, 2] - czc) <= h / two) & (lx > -l / two) & (lx < l / two) & (ly > -w / two) & (ly < w / two)
            hit = inside & ~found
            out[bi, hit] = t
            found |= inside
    return torch.from_numpy(out).to(points.device)


def heatmap_box_gather(raw, idx, bev_pos, K):
    """FD:606-629 (thin form) + FD:708-722: the (B, 6 * 10, H, W) output of a stage's task head, expanded task -> classes, cell
    offsets added, clipped, gathered at the stage's top proposals -> query_box (B, 10, k)."""
    B, _, H, W = raw.shape
    HW = H * W
    per_class = torch.stack([raw[:, 10 * t:10 * t + 10] for t in HEATMAP_TASK_OF_CLASS[:K]], 2).reshape(B, 10, K, HW).clone()
    per_class[:, :2] += bev_pos.int().float().transpose(1, 2)[:, :, None]
    for a, b_, lo, hi in HEATMAP_BOX_CLIPS:
        per_class[:, a:b_] = per_class[:, a:b_].clip(min=lo, max=hi)
    return per_class.view(B, 10, K * HW).gather(2, idx[:, None, :].expand(-1, 10, -1))


def box_class_mask(query_box, labels, bev_pos, K, cfg, margin=1.0, min_bev_dim=0.7):
    """FD:732-768, the box part of mask_heatmap_mode='boxcls': every BEV cell centre inside a (shrunk) query box takes the class
    of the first such query -> (B, K * HW) {0,1}."""
    B, HW = bev_pos.shape[:2]
    rot, dim, center, height, vel = query_box[:, 6:8], query_box[:, 3:6], query_box[:, 0:2], query_box[:, 2:3], query_box[:, 8:]
    std = decode_box(rot.clone(), dim.clone(), center.clone(), height.clone(), vel.clone(), cfg)
    std[..., 0] = std[..., 0].clip(min=-54.0, max=54.0)                   # FD:746-748: the nuScenes range, hard-coded
    std[..., 1] = std[..., 1].clip(min=-54.0, max=54.0)
    std[..., 3:5] = (std[..., 3:5] - margin).clip(min=min_bev_dim, max=10.0)
    std[..., 5] = 1000
    std[..., 2] = -100.0
    osf, (vx, vy), (px, py) = cfg.out_size_factor, cfg.voxel_size, cfg.pc_range[:2]
    pts = torch.stack([bev_pos[..., 0] * osf * vx + px, bev_pos[..., 1] * osf * vy + py, torch.zeros(B, HW)], -1)   # BC:46-52
    inside = points_in_boxes(pts, std[:, :, :7])
    cls = labels.gather(1, inside.clip(min=0).long())
    cls[inside == -1] = K
    sel = query_box.new_zeros(B, K + 1, HW)
    sel.scatter_(1, cls[:, None], torch.ones_like(cls[:, None], dtype=sel.dtype))
    return sel[:, :K].reshape(B, K * HW)


def mask_update(acc_masks, top_proposals, K, H, W, mode, ksize, small_classes, box_sel=None):
    """FD:725-782.  acc_masks (B, K*H*W) in {0,1}; returns the new acc_masks.  ``box_sel``: box_class_mask's result ('boxcls')."""
    B = acc_masks.shape[0]
    HW = H * W
    if mode == 'boxcls':                                                   # FD:732-770
        sel = acc_masks.new_zeros(B, K * HW)
        sel.scatter_(1, top_proposals, torch.ones_like(top_proposals, dtype=acc_masks.dtype))
        sel = (sel + box_sel > 0.1).float()
    elif mode == 'poscls':
        sel = acc_masks.new_zeros(B, K * HW)
        sel.scatter_(1, top_proposals, torch.ones_like(top_proposals, dtype=acc_masks.dtype))
    elif mode == 'pos':
        cell = top_proposals % HW
        sel = acc_masks.new_zeros(B, K, HW)
        sel.scatter_(2, cell[:, None, :].expand(-1, K, -1), acc_masks.new_ones(B, K, HW))
    else:  # FD:771-772
        sel = acc_masks.new_zeros(B, K * HW)
    sel = sel.reshape(B, K, H, W)
    dil = F.max_pool2d(sel, kernel_size=ksize, stride=1, padding=ksize // 2)
    for c in small_classes:
        if c < K:
            dil[:, c] = sel[:, c]
    return acc_masks * (1.0 - dil).view(B, -1)


def hip_stage(feat, logits, acc_masks, cfg, sd, box_raw=None):
    """One Hard-Instance-Probing stage, FD:631-634/662-666 + FD:670-706 + FD:725-782.

    feat   (B,C,H,W)  stage BEV map the query features are gathered from
    logits (B,K,H,W)  heatmap-head output for this stage
    Returns dict(idx, cls, cell, feat (B,C,k), pos (B,k,2), score (B,K,k), heat (B,K,HW)),
    new acc_masks.
    """
    B, K, H, W = logits.shape
    HW = H * W
    small = SMALL_CLASSES[cfg.dataset]
    heat = logits.sigmoid() * acc_masks.view(B, K, H, W)
    heat = local_max_nms(heat, cfg.nms_kernel_size, small).view(B, K, HW)
    idx = topk_deterministic(heat.view(B, -1), cfg.num_proposals)
    cls = idx // HW
    cell = idx % HW
    C = feat.shape[1]
    qf = feat.view(B, C, HW).gather(2, cell[:, None, :].expand(-1, C, -1))
    one_hot = F.one_hot(cls, num_classes=K).permute(0, 2, 1).float()
    qf = qf + F.conv1d(one_hot, sd['class_encoding.weight'], sd['class_encoding.bias'])  # FD:697-700
    bev_pos = create_2d_grid(H, W).repeat(B, 1, 1)
    qp = bev_pos.gather(1, cell[:, :, None].expand(-1, -1, 2))
    qs = heat.gather(2, cell[:, None, :].expand(-1, K, -1))
    qbox = box_sel = None
    if box_raw is not None:                                               # FD:708-722
        qbox = heatmap_box_gather(box_raw, idx, bev_pos, K)
        if cfg.mask_heatmap_mode == 'boxcls':
            box_sel = box_class_mask(qbox, cls, bev_pos, K, cfg)
    new_masks = mask_update(acc_masks, idx, K, H, W, cfg.mask_heatmap_mode, cfg.nms_kernel_size, small, box_sel)
    return dict(idx=idx, cls=cls, cell=cell, feat=qf, pos=qp, score=qs, heat=heat, box=qbox), new_masks


# --------------------------------------------------------------------------------------
# Deformable decoder (third-party arithmetic)
# --------------------------------------------------------------------------------------
def msda_core(value, spatial_shapes, sampling_locations, attention_weights):
    """A.3 core, Python form (mmcv ``multi_scale_deformable_attn_pytorch``).

    value (B,Nv,heads,Dh); spatial_shapes list[(H_l,W_l)]; sampling_locations
    (B,Nq,heads,L,P,2) in [0,1] (x,y); attention_weights (B,Nq,heads,L,P) -> (B,Nq,heads*Dh).
    """
    B, _, M, D = value.shape
    _, Nq, _, L, P, _ = sampling_locations.shape
    value_list = value.split([h * w for h, w in spatial_shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for l, (h, w) in enumerate(spatial_shapes):
        v = value_list[l].flatten(2).transpose(1, 2).reshape(B * M, D, h, w)
        g = grids[:, :, :, l].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros', align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(B * M, 1, Nq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(B, M * D, Nq)
    return out.transpose(1, 2).contiguous()


def msda_core_loops(value, spatial_shapes, sampling_locations, attention_weights):
    """A.3 core, CUDA form (``ms_deformable_im2col_gpu_kernel``) as explicit numpy loops:
    h_im = y*H - 0.5, w_im = x*W - 0.5, contributes iff -1 < h_im < H and -1 < w_im < W,
    each corner individually bounds-checked.  Small cases only."""
    v = value.numpy().astype(np.float64)
    loc = sampling_locations.numpy().astype(np.float64)
    aw = attention_weights.numpy().astype(np.float64)
    B, _, M, D = v.shape
    _, Nq, _, L, P, _ = loc.shape
    starts = np.cumsum([0] + [h * w for h, w in spatial_shapes])
    out = np.zeros((B, Nq, M, D))
    for b in range(B):
        for q in range(Nq):
            for m in range(M):
                for l, (H, W) in enumerate(spatial_shapes):
                    for p in range(P):
                        w_im = loc[b, q, m, l, p, 0] * W - 0.5
                        h_im = loc[b, q, m, l, p, 1] * H - 0.5
                        if not (h_im > -1 and w_im > -1 and h_im < H and w_im < W):
                            continue
                        h0, w0 = int(np.floor(h_im)), int(np.floor(w_im))
                        lh, lw = h_im - h0, w_im - w0
                        acc = np.zeros(D)
                        for dy, dx, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw),
                                           (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
                            y, x = h0 + dy, w0 + dx
                            if 0 <= y < H and 0 <= x < W:
                                acc += wt * v[b, starts[l] + y * W + x, m]
                        out[b, q, m] += aw[b, q, m, l, p] * acc
    return torch.from_numpy(out.reshape(B, Nq, M * D)).float()


# Test-harness hook (oracle/gen_golden.py:gen_train_step): callable (loc, spatial_shapes) -> loc applied to the sampling
# locations of every msda_module call, see desingularise_sampling().  None = the plain algorithm.
MSDA_LOC_HOOK = None


def desingularise_sampling(loc, spatial_shapes, tau=1e-3):
    """Bilinear interpolation is continuous in the sampling position but its DERIVATIVE with respect to the position jumps
    where a pixel coordinate (x * W - 0.5, y * H - 0.5) crosses an integer: the one-sided slopes v[n] - v[n-1] and
    v[n+1] - v[n] differ.  A sample within rounding of such a point gets one slope or the other depending on the last bit of
    the offset GEMM that produced it, so a recorded gradient (tests/golden/train_step_*.npz) would be a coin toss on another
    machine - one flipped sample is 3-4 % of the largest sampling-offset gradient entry of those fixtures, and with ~180 000
    sampling coordinates per step some always sit within a few ulps of a crossing.  The fixture therefore evaluates the step at
    DE-SINGULARISED locations: every coordinate closer than ``tau`` pixels to a crossing is moved to distance ``tau`` on its own
    side (<= 1e-3 px, a few dozen of the 180 000), the moved values are recorded, and the parity tests inject exactly those
    values (tests/train_step_util.py), so both sides differentiate the same function away from its kinks.
    loc (B, Nq, heads, L, P, 2) normalised -> (new loc with the gradient of ``loc`` (straight-through), flat indices int64,
    new values fp32 of the moved coordinates)."""
    new = loc.detach().clone()
    for l, (H, W) in enumerate(spatial_shapes):
        for axis, size in ((0, float(W)), (1, float(H))):
            t = new[..., l, :, axis] * size - 0.5
            n = torch.round(t)
            d = t - n
            risky = d.abs() < tau
            safe = (n + torch.where(d >= 0, torch.full_like(d, tau), torch.full_like(d, -tau)) + 0.5) / size
            new[..., l, :, axis] = torch.where(risky, safe, new[..., l, :, axis])
    idx = torch.nonzero(new.reshape(-1) != loc.detach().reshape(-1))[:, 0]
    return loc + (new - loc.detach()), idx, new.reshape(-1)[idx].clone()


def msda_module(query, value, identity, query_pos, reference_points, spatial_shapes, sd, p, heads, L, P, lowp=False):
    """A.3 ``MultiScaleDeformableAttention.forward`` (batch_first=False, eval: dropout off).
    query/identity/query_pos (Nq,B,C), value (Nv,B,C), reference_points (B,Nq,1,2)."""
    if identity is None:
        identity = query
    if query_pos is not None:
        query = query + query_pos
    query = query.permute(1, 0, 2)
    value = value.permute(1, 0, 2)
    B, Nq, C = query.shape
    Nv = value.shape[1]
    assert sum(h * w for h, w in spatial_shapes) == Nv
    value = lin(value, sd[p + 'value_proj.weight'], sd[p + 'value_proj.bias'], lowp).view(B, Nv, heads, -1)
    off = F.linear(query, sd[p + 'sampling_offsets.weight'], sd[p + 'sampling_offsets.bias'])   # (offsets / logits: fp32 in either mode)
    off = off.view(B, Nq, heads, L, P, 2)
    aw = F.linear(query, sd[p + 'attention_weights.weight'], sd[p + 'attention_weights.bias'])
    aw = aw.view(B, Nq, heads, L * P).softmax(-1).view(B, Nq, heads, L, P)
    normalizer = torch.tensor([[w, h] for h, w in spatial_shapes], dtype=query.dtype)
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    if MSDA_LOC_HOOK is not None:
        loc = MSDA_LOC_HOOK(loc, spatial_shapes)
    out = msda_core(value, spatial_shapes, loc, aw)
    out = lin(out, sd[p + 'output_proj.weight'], sd[p + 'output_proj.bias'], lowp)
    return out.permute(1, 0, 2) + identity


def mha_module(query, query_pos, sd, p, heads, attn_mask=None, lowp=False):
    """A.2 mmcv ``MultiheadAttention`` as self-attention: q = k = x + pos, v = x,
    torch ``nn.MultiheadAttention`` semantics, residual on the pre-pos query."""
    identity = query
    qk = query + query_pos if query_pos is not None else query
    C = query.shape[-1]
    if lowp:
        # the same function with its four projections on lin(lowp=True); softmax(q k^T / sqrt(Dh)) v stays fp32
        assert attn_mask is None
        w, b = sd[p + 'attn.in_proj_weight'], sd[p + 'attn.in_proj_bias']
        N, B = query.shape[:2]
        q = lin(qk, w[:C], b[:C], True).view(N, B * heads, C // heads).transpose(0, 1)
        k = lin(qk, w[C:2 * C], b[C:2 * C], True).view(N, B * heads, C // heads).transpose(0, 1)
        v = lin(query, w[2 * C:], b[2 * C:], True).view(N, B * heads, C // heads).transpose(0, 1)
        a = torch.softmax(torch.bmm(q, k.transpose(1, 2)) / math.sqrt(C // heads), -1)
        o = torch.bmm(a, v).transpose(0, 1).reshape(N, B, C)
        return identity + lin(o, sd[p + 'attn.out_proj.weight'], sd[p + 'attn.out_proj.bias'], True)
    out = F.multi_head_attention_forward(
        qk, qk, query, C, heads,
        sd[p + 'attn.in_proj_weight'], sd[p + 'attn.in_proj_bias'], None, None, False, 0.0,
        sd[p + 'attn.out_proj.weight'], sd[p + 'attn.out_proj.bias'],
        training=False, need_weights=False, attn_mask=attn_mask)[0]
    return identity + out


def ffn_module(x, sd, p, lowp=False):
    """A.2 mmcv ``FFN`` (num_fcs=2, ReLU, add_identity)."""
    y = lin(x, sd[p + 'layers.0.0.weight'], sd[p + 'layers.0.0.bias'], lowp, relu=True)
    y = lin(y, sd[p + 'layers.1.weight'], sd[p + 'layers.1.bias'], lowp)
    return x + y


def layer_norm(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + 'weight'], sd[p + 'bias'], LN_EPS)


def decoder_layer(query, value, query_pos, reference_points_input, spatial_shapes, sd, p, cfg, attn_mask=None):
    """A.2 ``DetrTransformerDecoderLayer`` with operation_order
    ('self_attn','norm','cross_attn','norm','ffn','norm') (FocalFormer3D_L.py:312-313)."""
    lowp = getattr(cfg, 'gemm_dtype', 'f32') == 'bf16'
    x = mha_module(query, query_pos, sd, p + 'attentions.0.', cfg.num_heads, attn_mask, lowp)
    x = layer_norm(x, sd, p + 'norms.0.')
    x = msda_module(x, value, None, query_pos, reference_points_input, spatial_shapes, sd,
                    p + 'attentions.1.', cfg.num_heads, cfg.num_levels, cfg.num_points, lowp)
    x = layer_norm(x, sd, p + 'norms.1.')
    x = ffn_module(x, sd, p + 'ffns.0.', lowp)
    x = layer_norm(x, sd, p + 'norms.2.')
    return x


def deformable_decoder(query, value, query_pos, reference_points, spatial_shapes, valid_ratios, sd, p, cfg,
                       attn_mask=None, taps=None):
    """A.1 ``DeformableDetrTransformerDecoder.forward`` (reg_branches None, return_intermediate
    False) as called at FD:927-933.  query/query_pos (Nq,B,C), value (Nv,B,C),
    reference_points (B,Nq,2)."""
    out = query
    for l in range(cfg.num_layers):
        ref_in = reference_points[:, :, None] * valid_ratios[:, None]
        out = decoder_layer(out, value, query_pos, ref_in, spatial_shapes, sd, f'{p}layers.{l}.', cfg, attn_mask)
        if taps is not None:
            taps.append(out)
    return out, reference_points


# --------------------------------------------------------------------------------------
# RoI grid features, prediction heads, box coder
# --------------------------------------------------------------------------------------
def decode_box(rot, dim, center, height, vel, cfg):
    """BC:54-69.  Inputs (B,n,Nq) -> (B,Nq,7[+2]) metric (x,y,z_bottom,w,l,h,yaw[,vx,vy])."""
    center = center.clone()
    center[:, 0] = center[:, 0] * cfg.out_size_factor * cfg.voxel_size[0] + cfg.pc_range[0]
    center[:, 1] = center[:, 1] * cfg.out_size_factor * cfg.voxel_size[1] + cfg.pc_range[1]
    dim = dim.exp()
    height = height - dim[:, 2:3] * 0.5
    yaw = torch.atan2(rot[:, 0:1], rot[:, 1:2])
    parts = [center, height, dim, yaw] + ([] if vel is None else [vel])
    return torch.cat(parts, dim=1).permute(0, 2, 1)


def roi_grid_points(query_box, expand, g, cfg):
    """FD:891-909 + FD:1655-1664 + A.5 -> normalised sampling grid (B,Nq,g*g,2) in [-2,2]."""
    B, _, Nq = query_box.shape
    rot, dim, center, height, vel = (query_box[:, 6:8], query_box[:, 3:6], query_box[:, 0:2],
                                     query_box[:, 2:3], query_box[:, 8:])
    std = decode_box(rot, dim * expand, center, height, vel if vel.shape[1] else None, cfg)
    std = std.reshape(B * Nq, -1)
    idx = torch.ones(g, g).nonzero().float()                      # (g*g, 2), first index slow
    size = std[:, 3:5]
    pts = (idx[None] + 0.5) / g * size[:, None] - size[:, None] / 2  # FD:1662-1663
    yaw = std[:, 6]
    c, s = torch.cos(yaw)[:, None], torch.sin(yaw)[:, None]
    x = pts[..., 0] * c + pts[..., 1] * s                          # A.5 rotation_3d_in_axis, axis=2
    y = -pts[..., 0] * s + pts[..., 1] * c
    pts = torch.stack([x, y], -1) + std[:, None, :2]
    pts = pts.view(B, Nq, g * g, 2)
    lo = torch.tensor(ROI_PC_RANGE[cfg.dataset][:2])
    hi = torch.tensor(ROI_PC_RANGE[cfg.dataset][2:])
    pts = (pts - lo) / (hi - lo)
    return (pts * 2.0 - 1.0).clip(min=-2.0, max=2.0)


def roi_sample(levels, grid):
    """FD:911-919: bilinear zero-padded align_corners=False sampling of every pyramid level,
    output (B*Nq, L*C*g*g) with column order [level][channel][grid point]."""
    B, Nq = grid.shape[:2]
    feats = [F.grid_sample(f, grid, mode='bilinear', padding_mode='zeros', align_corners=False) for f in levels]
    roi = torch.cat(feats, dim=1)                                   # (B, L*C, Nq, g*g)
    return roi.permute(0, 2, 1, 3).reshape(B * Nq, -1)


def roi_mlp(x, sd, p='roi_mlp.', lowp=False):
    """FD:186-200: 3 x (Linear no-bias, BN1d eval, ReLU[, Dropout eval]).  ``lowp`` (see lin()): the inference-time algebra
    first - BatchNorm folded into the layer, W' = W * gamma / sqrt(var + eps), b' = beta - mean * gamma / sqrt(var + eps), in fp32 -
    then the three layers as bf16 GEMMs chained through bf16 activations, the sampled RoI matrix rounded to bf16 on entry."""
    idx = sorted(int(k[len(p):].split('.')[0]) for k in sd
                 if k.startswith(p) and k.endswith('.weight') and sd[k].dim() == 2)
    if lowp:
        for i in idx:
            scale = sd[f'{p}{i + 1}.weight'] / torch.sqrt(sd[f'{p}{i + 1}.running_var'] + BN_EPS)
            x = lin(x, sd[f'{p}{i}.weight'] * scale[:, None], sd[f'{p}{i + 1}.bias'] - sd[f'{p}{i + 1}.running_mean'] * scale,
                    True, relu=True)
        return x
    for i in idx:
        x = F.linear(x, sd[f'{p}{i}.weight'])
        x = F.batch_norm(x, sd[f'{p}{i + 1}.running_mean'], sd[f'{p}{i + 1}.running_var'],
                         sd[f'{p}{i + 1}.weight'], sd[f'{p}{i + 1}.bias'], False, 0.0, BN_EPS)
        x = F.relu(x)
    return x


def prediction_heads(x, sd, p, head_names):
    """DU:495-578 ``FFN``: per head Conv1d(C->64,k1,no bias)+BN1d+ReLU, Conv1d(64->n,k1,bias)."""
    out = {}
    for name in head_names:
        q = f'{p}{name}.'
        y = F.conv1d(x, sd[q + '0.conv.weight'])
        y = F.batch_norm(y, sd[q + '0.bn.running_mean'], sd[q + '0.bn.running_var'],
                         sd[q + '0.bn.weight'], sd[q + '0.bn.bias'], False, 0.0, BN_EPS)
        y = F.relu(y)
        out[name] = F.conv1d(y, sd[q + '1.weight'], sd[q + '1.bias'])
    return out


def bbox_decode(heatmap, rot, dim, center, height, vel, cfg):
    """BC:71-158 ``decode(filter=True)``.  Returns per-sample dicts(bboxes, scores, labels) and
    the un-filtered (boxes, scores, labels, keep-mask) tensors."""
    labels = heatmap.max(1).indices
    scores = heatmap.max(1).values
    boxes = decode_box(rot, dim, center, height, vel, cfg)
    pcr = torch.tensor(cfg.post_center_range)
    mask = (boxes[..., :3] >= pcr[:3]).all(2) & (boxes[..., :3] <= pcr[3:]).all(2)
    if cfg.score_threshold:                                          # BC:140-141 (0.0 is falsy)
        mask &= scores > cfg.score_threshold
    dicts = [dict(bboxes=boxes[i, mask[i]], scores=scores[i, mask[i]], labels=labels[i, mask[i]])
             for i in range(heatmap.shape[0])]
    return dicts, (boxes, scores, labels, mask)


# --------------------------------------------------------------------------------------
# FocalDecoder.forward (inference) and get_bboxes
# --------------------------------------------------------------------------------------
def focal_decoder_forward(sd, cfg, pts_inputs, taps=None):
    """FD:522-992, eval mode.  ``pts_inputs`` = [pts_feat_conv, stage maps (list | tensor)];
    the list is NOT mutated (the reference pops/inserts, FD:528,592).
    Returns (result dict as FD:960-992, aux dict with query_labels/num_proposals)."""
    sd = {k: v for k, v in sd.items()}
    lidar_feat = pts_inputs[0]
    B, C, H, W = lidar_feat.shape
    HW = H * W
    K = cfg.num_classes
    small = SMALL_CLASSES[cfg.dataset]
    stage_list = list(pts_inputs[1]) if isinstance(pts_inputs[1], (list, tuple)) else pts_inputs[1]
    extra = None
    if cfg.extra_feat:
        extra = stage_list[-1]
        stage_list = stage_list[:-1]
    bev_pos = create_2d_grid(H, W).repeat(B, 1, 1)

    heatmap_train = []
    masks_out = []
    if not cfg.num_stages:
        # ---- single-stage branch, FD:539-586 (DeformFormer3D)
        dense = heatmap_head(lidar_feat, sd, 'heatmap_head.')
        if cfg.input_img or cfg.iterbev_wo_img:
            new_feat = stage_list[-1] if isinstance(stage_list, list) else stage_list
            dense_img = heatmap_head(new_feat.view(lidar_feat.shape), sd, 'heatmap_head_img.')
            heat = (dense.sigmoid() + dense_img.sigmoid()) / 2
            heatmap_train = [dense, dense_img]
        else:
            new_feat = lidar_feat
            heat = dense.sigmoid()
            heatmap_train = dense
        heat = local_max_nms(heat, cfg.nms_kernel_size, small).view(B, K, HW)
        idx = topk_deterministic(heat.view(B, -1), cfg.num_proposals)
        cls, cell = idx // HW, idx % HW
        qf = new_feat.reshape(B, C, HW).gather(2, cell[:, None, :].expand(-1, C, -1))
        one_hot = F.one_hot(cls, num_classes=K).permute(0, 2, 1).float()
        qf = qf + F.conv1d(one_hot, sd['class_encoding.weight'], sd['class_encoding.bias'])
        query_pos = bev_pos.gather(1, cell[:, :, None].expand(-1, -1, 2))
        query_score = heat.gather(2, cell[:, None, :].expand(-1, K, -1))
        query_feat, query_labels = qf, cls
        num_proposals = cfg.num_proposals
        pyramid_src = flat_src = new_feat
        stage_taps = [dict(idx=idx, heat=heat)]
    else:
        # ---- multi-stage Hard Instance Probing, FD:587-791
        dense0 = heatmap_head(lidar_feat, sd, 'heatmap_head.')
        feats = list(stage_list)
        if cfg.reuse_first_heatmap:
            feats.insert(0, lidar_feat)
        acc = torch.ones(B, K * HW)
        outs = []
        stage_taps = []
        bev_preds = []
        for i in range(cfg.num_stages):
            if i == 0 and cfg.reuse_first_heatmap:
                logits = dense0
                heatmap_train.append(dense0)
                masks_out.append(acc.view(B, K, H, W).clone())
            else:
                logits = heatmap_head(feats[i], sd, f'heatmap_head_img.{i}.')
                if i == 0:
                    heatmap_train.append(dense0)
                    masks_out.append(acc.view(B, K, H, W).clone())
                heatmap_train.append(logits)
                masks_out.append(acc.view(B, K, H, W).clone())
            box_raw = None
            if cfg.heatmap_box:                                           # FD:606-629 / 641-660, thin form: conv + BN + ReLU, conv -> 6 x 10
                assert cfg.thin_heatmap_box and cfg.dataset == 'nuScenes', 'heatmap_box: only the thin form (FD:260-281) is restated'
                box_raw = heatmap_head(feats[i], sd, f'multi_stage_task_heads.{i}.')
                bev_preds.append(box_raw)
            st, acc = hip_stage(feats[i], logits, acc, cfg, sd, box_raw)
            outs.append(st)
            stage_taps.append(dict(idx=st['idx'], heat=st['heat'], acc=acc.clone()))
        query_labels = torch.cat([o['cls'] for o in outs], 1)
        query_feat = torch.cat([o['feat'] for o in outs], 2)
        query_pos = torch.cat([o['pos'] for o in outs], 1)
        query_score = torch.cat([o['score'] for o in outs], 2)
        if cfg.heatmap_box:
            query_box0 = torch.cat([o['box'] for o in outs], 2)           # FD:788-789
        num_proposals = cfg.num_proposals * cfg.num_stages
        pyramid_src = extra if cfg.extra_feat else feats[-1]
        flat_src = feats[-1]                                            # FD:670 (non-multiscale value source)
    if taps is not None:
        taps['stages'] = stage_taps
        taps['query_feat0'] = query_feat.clone()
        taps['query_pos0'] = query_pos.clone()

    # ---- BEV pyramid, FD:810-823
    if cfg.multiscale:
        levels = [pyramid_src]
        levels.append(conv_module_2d(levels[-1], sd, 'dconv.', stride=2))
        levels.append(conv_module_2d(levels[-1], sd, 'dconv2.', stride=2))
        bev_pos_all = torch.cat([bev_pos,
                                 create_2d_grid(H // 2, H // 2).repeat(B, 1, 1) * 2,
                                 create_2d_grid(H // 4, H // 4).repeat(B, 1, 1) * 4], 1)  # FD:534-535,847
    else:
        levels = [flat_src]
        bev_pos_all = bev_pos
    flat = torch.cat([f.flatten(2, 3) for f in levels], -1)            # (B,C,Nv)
    spatial_shapes = [tuple(f.shape[2:]) for f in levels]
    wh = torch.tensor([float(spatial_shapes[0][1]), float(spatial_shapes[0][0])])  # flip(spatial_shapes[:1]) FD:869

    head_names = list(cfg.common_heads.keys()) + ['heatmap']
    ret = []
    query_box = query_box0 if (cfg.num_stages and cfg.heatmap_box) else None
    for s in range(cfg.num_decoder_layers):
        reference_points = query_pos / wh                               # FD:869
        qpe = mlp(gen_sineembed_for_position(reference_points), sd, f'pos_embed_learned.{s}.')
        if cfg.bevpos:
            bpe = mlp(gen_sineembed_for_position(bev_pos_all / wh), sd, f'pos_embed_learned.{s}.')
            value = flat + bpe.transpose(1, 2)                          # FD:883-886
        else:
            value = flat
        if cfg.roi_feats and query_box is not None:                     # FD:890-922
            grid = roi_grid_points(query_box, cfg.roi_expand_ratio[s], cfg.roi_feats, cfg)
            roi = roi_sample(levels, grid)
            if taps is not None:
                taps.setdefault('roi_grid', []).append(grid)
                taps.setdefault('roi_mat', []).append(roi)
            roi = roi_mlp(roi, sd, lowp=cfg.gemm_dtype == 'bf16')
            query_feat = query_feat + roi.view(B, num_proposals, C).transpose(1, 2)
        layer_taps = [] if taps is not None else None
        x, reference_points = deformable_decoder(
            query_feat.permute(2, 0, 1), value.permute(2, 0, 1), qpe.permute(1, 0, 2), reference_points,
            spatial_shapes, torch.ones(B, 1, 2), sd, f'decoder.{s}.', cfg, taps=layer_taps)
        if taps is not None:
            taps.setdefault('decoder_layers', []).append(layer_taps)
        query_feat = x.permute(1, 2, 0)
        query_pos = reference_points * wh                               # FD:936
        res = prediction_heads(query_feat, sd, f'prediction_heads.{s}.', head_names)
        if cfg.classaware_reg:                                          # FD:940-943
            for k in ('center', 'height', 'dim', 'rot'):
                r = res[k].view(B, K, -1, num_proposals)
                res[k] = r.gather(1, query_labels[:, None, None, :].expand(-1, -1, r.shape[2], -1)
                                  .clip(0, K - 1))[:, 0]
        res['center'] = res['center'] + query_pos.permute(0, 2, 1)       # FD:945
        query_pos = res['center'].clone().permute(0, 2, 1)
        if cfg.roi_based_reg and query_box is not None:                 # FD:949-951
            res['dim'] = torch.cat([res['dim'][:, :2] + query_box[:, 3:5], res['dim'][:, 2:]], 1)
            res['rot'] = res['rot'] + query_box[:, 6:8]
        parts = [res['center'], res['height'], res['dim'], res['rot']] + ([res['vel']] if 'vel' in res else [])
        query_box = torch.cat(parts, 1)
        ret.append(res)

    out = {k: torch.cat([r[k] for r in ret], -1) for k in ret[0]}
    out['query_heatmap_score'] = query_score
    out['dense_heatmap'] = heatmap_train
    if cfg.num_stages:
        out['multistage_masks'] = masks_out
    if cfg.heatmap_box:                                                  # FD:988-991 (bev preds: one (B,60,H,W) tensor per stage here)
        out['multistage_bev_preds'] = bev_preds
        out['query_pos'] = query_pos
        out['query_box'] = query_box
    aux = dict(query_labels=query_labels, num_proposals=num_proposals)
    return out, aux


def focal_decoder_get_bboxes(out, aux, cfg):
    """FD:1313-1413 with test_cfg.nms_type=None (every shipped config), batch generalised:
    returns one (boxes (n,9|7), scores (n,), labels int (n,)) triple per sample."""
    n = aux['num_proposals']
    K = cfg.num_classes
    score = out['heatmap'][..., -n:].sigmoid()
    one_hot = F.one_hot(aux['query_labels'], num_classes=K).permute(0, 2, 1)
    score = score * out['query_heatmap_score'] * one_hot
    vel = out['vel'][..., -n:].clone() if 'vel' in out else None
    dicts, raw = bbox_decode(score, out['rot'][..., -n:].clone(), out['dim'][..., -n:].clone(),
                             out['center'][..., -n:].clone(), out['height'][..., -n:].clone(), vel, cfg)
    res = []
    for d in dicts:
        b, s, l = d['bboxes'], d['scores'], d['labels']
        if len(b) > 200:                                                 # FD:1395-1400
            inds = s.argsort(descending=True)[:200]
            b, s, l = b[inds], s[inds], l[inds]
        res.append((b, s, l.int()))
    return res, raw


# --------------------------------------------------------------------------------------
# mmdet3d v0.17.1 apply_3d_transformation (A.5; un-vendored: restated from the published algorithm, "parity unpinned" by
# execution - pinned by the forward / reverse round trip and by a hand-computed case in tests/test_host_cpu.py)
# --------------------------------------------------------------------------------------
def apply_3d_transformation(pcd, img_meta, reverse=False):
    """mmdet3d/models/fusion_layers/coord_transform.py:apply_3d_transformation(pcd, 'LIDAR', img_meta, reverse): the recorded
    point-cloud augmentation flow applied step by step (BasePoints.rotate = points @ matrix, scale, translate, LiDARPoints.flip:
    'horizontal' negates y, 'vertical' negates x), reversed = the inverse steps in reverse order."""
    dt = pcd.dtype
    rot = torch.as_tensor(img_meta['pcd_rotation'], dtype=dt) if 'pcd_rotation' in img_meta else torch.eye(3, dtype=dt)
    scale = img_meta.get('pcd_scale_factor', 1.0)
    trans = torch.as_tensor(img_meta['pcd_trans'], dtype=dt) if 'pcd_trans' in img_meta else torch.zeros(3, dtype=dt)
    flow = list(img_meta.get('transformation_3d_flow', []))
    pcd = pcd.clone()
    if reverse:
        rot, scale, trans, flow = rot.inverse(), 1.0 / scale, -trans, flow[::-1]
    for op in flow:
        if op == 'T':
            pcd = pcd + trans
        elif op == 'S':
            pcd = pcd * scale
        elif op == 'R':
            pcd = pcd @ rot
        elif op == 'HF':
            if img_meta.get('pcd_horizontal_flip', False):
                pcd = pcd * pcd.new_tensor([1.0, -1.0, 1.0])
        elif op == 'VF':
            if img_meta.get('pcd_vertical_flip', False):
                pcd = pcd * pcd.new_tensor([-1.0, 1.0, 1.0])
        else:
            raise AssertionError(op)
    return pcd


# --------------------------------------------------------------------------------------
# I2P camera-projection sampler
# --------------------------------------------------------------------------------------
def create_3d_grid(x_size, y_size, z_size, dtype=torch.float32):
    """EU:174-182: flat index = (i*y_size + j)*z_size + k over the three linspaces; columns are
    (k, j, i) + 0.5 - the caller passes (Z, H, W) so columns are (x, y, z)."""
    a, b, c = torch.meshgrid(torch.linspace(0, x_size - 1, x_size, dtype=dtype), torch.linspace(0, y_size - 1, y_size, dtype=dtype),
                             torch.linspace(0, z_size - 1, z_size, dtype=dtype), indexing='ij')
    return torch.stack([c + 0.5, b + 0.5, a + 0.5], 0).view(1, 3, -1).permute(0, 2, 1)


def i2p_project(lidar2img, H, W, Z, input_shape, img_aug=None, return_depth=False, img_meta=None):
    """EU:210-242 for one sample: pillar-grid points -> per-camera normalised image coords.
    lidar2img (Ncam,4,4); returns xy (Ncam, Z*H*W, 2) in grid_sample convention, mask (Ncam, Z*H*W).  The arithmetic runs in
    ``lidar2img.dtype`` (float64 = the exact-arithmetic yardstick of the conditioning tests)."""
    dt = lidar2img.dtype
    pcr = torch.tensor([-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], dtype=dt)
    shape = [W, H, Z]
    grid = create_3d_grid(*shape[::-1], dtype=dt) / torch.tensor(shape, dtype=dt)
    grid = (grid * (pcr[3:] - pcr[:3]) + pcr[:3]).squeeze(0)
    if img_meta is not None:                                                      # EU:222: undo the point-cloud augmentation
        grid = apply_3d_transformation(grid, img_meta, reverse=True)
    pts = torch.cat([grid, torch.ones_like(grid[:, :1])], -1)[None, :, :, None]   # (1,N,4,1)
    cam = torch.matmul(lidar2img[:, None], pts).squeeze(-1)                       # (Ncam,N,4)
    eps = 1e-5
    mask = cam[..., 2:3] > eps
    xy = cam[..., 0:2] / torch.maximum(cam[..., 2:3], torch.ones_like(cam[..., 2:3]) * eps)
    if img_aug is not None:                                                        # EU:230-233
        post_rots, post_trans = img_aug[..., :3, :3], img_aug[..., :3, 3]
        xy1 = torch.cat([xy, xy.new_ones(*xy.shape[:-1], 1)], -1)
        n = xy.shape[0]
        xy = (post_rots.view(n, 1, 3, 3).matmul(xy1.unsqueeze(-1)).squeeze(-1) + post_trans.view(n, 1, 3))[..., :2]
    xy = torch.stack([xy[..., 0] / input_shape[1], xy[..., 1] / input_shape[0]], -1)
    xy = (xy - 0.5) * 2
    mask = mask & (xy[..., 0:1] > -1.0) & (xy[..., 0:1] < 1.0) & (xy[..., 1:2] > -1.0) & (xy[..., 1:2] < 1.0)
    if return_depth:                     # (tests: distance of every sample to the visibility decisions above)
        return xy, mask[..., 0], cam[..., 2]
    return xy, mask[..., 0]


def i2p_forward(sd, lidar_feat, img_feat, lidar2img, input_shape, Z, img_aug=None, p='learnedAlign.', taps=None, img_metas=None):
    """EU:194-261 ``I2P.forward`` (eval; pcd augmentation undo = identity at test time, A.5).
    lidar_feat (B,C,H,W); img_feat (B,Ncam,Ci,Hi,Wi); lidar2img (B,Ncam,4,4);
    img_aug (B,Ncam,4,4) or None."""
    B, C, H, W = lidar_feat.shape
    Ci = img_feat.shape[2]
    out = torch.zeros_like(lidar_feat)
    for b in range(B):
        xy, mask = i2p_project(lidar2img[b].to(lidar_feat.dtype), H, W, Z, input_shape, None if img_aug is None else img_aug[b],
                               img_meta=None if img_metas is None else img_metas[b])
        ncam = xy.shape[0]
        sampled = F.grid_sample(img_feat[b], xy.unsqueeze(-2), mode='bilinear', padding_mode='zeros',
                                align_corners=False).squeeze(-1)           # (Ncam,Ci,N)
        m = mask.view(ncam, 1, Z, H, W).to(lidar_feat.dtype)
        sampled = sampled.view(ncam, Ci, Z, H, W)
        red = (sampled * m).sum(0) / (m.sum(0) + 1e-10)                     # (Ci,Z,H,W)
        red = red.flatten(2, 3).transpose(0, 2)                             # (HW,Z,Ci)
        kmask = (m[:, 0].sum(0) > 0).view(Z, H * W).t()                     # (HW,Z)
        Q = lidar_feat[b].flatten(1, 2).t().unsqueeze(1)                    # (HW,1,C)
        valid = kmask.sum(1) > 0
        attn = lidar_feat.new_zeros(H * W, 1, C)
        if valid.any():
            if (p + 'in_proj_weight') in sd:
                kw = dict(in_proj_weight=sd[p + 'in_proj_weight'], use_separate_proj_weight=False)
            else:
                kw = dict(in_proj_weight=None, use_separate_proj_weight=True, q_proj_weight=sd[p + 'q_proj_weight'],
                          k_proj_weight=sd[p + 'k_proj_weight'], v_proj_weight=sd[p + 'v_proj_weight'])
            q, k = Q[valid].transpose(0, 1), red[valid].transpose(0, 1)      # seq-first
            o = F.multi_head_attention_forward(
                q, k, k, C, 1, in_proj_bias=sd[p + 'in_proj_bias'], bias_k=None, bias_v=None, add_zero_attn=False,
                dropout_p=0.0, out_proj_weight=sd[p + 'out_proj.weight'], out_proj_bias=sd[p + 'out_proj.bias'],
                training=False, need_weights=False, attn_mask=(~kmask[valid])[:, None, :], **kw)[0]
            attn[valid] = o.transpose(0, 1)
        out[b] = attn.squeeze(1).t().reshape(C, H, W)
        if taps is not None:
            taps.setdefault('xy', []).append(xy)
            taps.setdefault('mask', []).append(mask)
            taps.setdefault('reduced', []).append(red)
    return out


# --------------------------------------------------------------------------------------
# Local context attention (neck, iterbev='bevfusion'): reference CUDA extension
# projects/mmdet3d_plugin/models/utils/ops/locatt_ops restated on CPU.  "Parity unpinned by
# execution": the extension is CUDA-only (locatt_ops/__init__.py:11 asserts CUDA) and cannot be
# built or run here; restated from its source kernels.cuh (which IS under /root/reference).
# --------------------------------------------------------------------------------------
def _window_unfold(x, kH, kW):
    """(B,C,H,W) -> (B,C,kH*kW,H,W): tap k = dy*kW+dx reads x[..., h+dy-kH//2, w+dx-kW//2], zero outside."""
    B, C, H, W = x.shape
    return F.unfold(x, (kH, kW), padding=(kH // 2, kW // 2)).view(B, C, kH * kW, H, W)


def locatt_similar(x_ori, x_loc, kH, kW):
    """kernels.cuh:4-42 ``cc2k`` (driven per sample by similar.cu:3-38): y (B,H,W,kH*kW); window
    positions outside the map keep 0.  The reference accumulates in double and rounds to float."""
    u = _window_unfold(x_loc.double(), kH, kW)
    return (x_ori.double()[:, :, None] * u).sum(1).permute(0, 2, 3, 1).to(x_ori.dtype).contiguous()


def locatt_weighting(x_ori, x_weight, kH, kW):
    """kernels.cuh:44-80 ``ck2c_ori``: y[b,c,h,w] = sum_k x_ori[b,c,h+dy,w+dx] * w[b,h,w,k]."""
    u = _window_unfold(x_ori.double(), kH, kW)
    return (u * x_weight.double().permute(0, 3, 1, 2)[:, None]).sum(2).to(x_ori.dtype)


def conv_bn_relu_1x1(x, sd, p, relu=True):
    """EU:10-33 ``ConvBNReLU`` with kernel 1 (bias='auto' -> no conv bias under BN), eval mode."""
    y = F.conv2d(x, sd[p + 'conv.weight'], sd.get(p + 'conv.bias'))
    y = F.batch_norm(y, sd[p + 'bn.running_mean'], sd[p + 'bn.running_var'], sd.get(p + 'bn.weight'),
                     sd.get(p + 'bn.bias'), False, 0.0, BN_EPS)
    return F.relu(y) if relu else y


def local_context_attention(sd, target, source, k, p=''):
    """EU:109-163 ``LocalContextAttentionBlock.forward``."""
    q = conv_bn_relu_1x1(conv_bn_relu_1x1(target, sd, p + 'query_project.0.'), sd, p + 'query_project.1.')
    key = conv_bn_relu_1x1(conv_bn_relu_1x1(source, sd, p + 'key_project.0.'), sd, p + 'key_project.1.')
    val = conv_bn_relu_1x1(source, sd, p + 'value_project.')
    w = locatt_similar(q, key, k, k)
    w = F.softmax(w / math.sqrt(key.size(1)), -1)
    return locatt_weighting(val, w, k, k)


# --------------------------------------------------------------------------------------
# LSS pillar pooling: reference CUDA extension models/utils/ops/bev_pool restated on CPU
# (bev_pool_op.py:81-97 + bev_pool_cuda.cu:20-42).  CUDA-only in the reference -> restated from source.
# --------------------------------------------------------------------------------------
def bev_pool(feats, coords, B, D, H, W):
    """feats (n,c), coords (n,4) int (x, y, z, b) -> (B, c, D, H, W): sum of the features of all points of a cell."""
    c = feats.shape[1]
    out = torch.zeros(B, D, H, W, c, dtype=torch.float64)
    flat = ((coords[:, 3].long() * D + coords[:, 2].long()) * H + coords[:, 0].long()) * W + coords[:, 1].long()
    out.view(-1, c).index_add_(0, flat, feats.double())
    return out.permute(0, 4, 1, 2, 3).float().contiguous()


# --------------------------------------------------------------------------------------
# get_bboxes with test_cfg.nms_type == 'circle' (FD:1352-1393).  `circle_nms` is mmdet3d v0.17.1
# (mmdet3d/core/post_processing/box3d_nms.py, un-vendored third party) restated from its published
# algorithm - "parity unpinned"; the task tables and the surrounding logic are the reference's (FD:1333-1393).
# --------------------------------------------------------------------------------------
NMS_TASKS = {'nuScenes': [([0, 1, 2, 3, 4, 5, 6, 7], -1.0), ([8], 0.175), ([9], 0.175)],
             'Waymo': [([0], 0.7), ([1], 0.7), ([2], 0.7)]}


def circle_nms(dets, thresh, post_max_size=83):
    """dets (n,3) numpy [x, y, score] -> kept indices (greedy by descending score, squared distance <= thresh)."""
    order = np.argsort(-dets[:, 2], kind='stable')
    suppressed = np.zeros(len(dets), dtype=bool)
    keep = []
    for a in range(len(order)):
        i = order[a]
        if suppressed[i]:
            continue
        keep.append(i)
        for bb in range(a + 1, len(order)):
            jj = order[bb]
            if not suppressed[jj] and (dets[i, 0] - dets[jj, 0]) ** 2 + (dets[i, 1] - dets[jj, 1]) ** 2 <= thresh:
                suppressed[jj] = True
    return keep[:post_max_size]


def get_bboxes_circle_nms(dicts, cfg):
    """FD:1347-1393 on the per-sample dicts of bbox_decode(filter=True)."""
    res = []
    for d in dicts:
        b, s, l = d['bboxes'], d['scores'], d['labels']
        keep_mask = torch.zeros_like(s, dtype=torch.bool)
        for idx, radius in NMS_TASKS[cfg.dataset]:
            task_mask = torch.zeros_like(s, dtype=torch.bool)
            for c in idx:
                task_mask |= l == c
            if radius > 0:
                dets = torch.cat([b[task_mask][:, :2], s[task_mask][:, None]], 1).numpy().astype(np.float32)
                kept = torch.tensor(circle_nms(dets, np.float32(radius)), dtype=torch.long)
            else:
                kept = torch.arange(int(task_mask.sum()))
            if kept.numel():
                keep_mask[torch.where(task_mask)[0][kept]] = True
        b, s, l = b[keep_mask], s[keep_mask], l[keep_mask]
        if len(b) > 200:
            inds = s.argsort(descending=True)[:200]
            b, s, l = b[inds], s[inds], l[inds]
        res.append((b, s, l.int()))
    return res


# --------------------------------------------------------------------------------------
# Rotated BEV IoU, rotated NMS and TTA merging.  `box_overlap` / `iou_bev` / `nms_gpu` / `boxes_iou_bev` are mmdet3d 0.17.1
# iou3d ops (mmdet3d/ops/iou3d/src/iou3d_kernel.cu, iou3d_utils.py; un-vendored third party, restated from the published
# algorithm - "parity unpinned"); the callers are the reference's: FD:1369-1383 and
# projects/mmdet3d_plugin/core/post_processing/merge_augs.py:113-184.  Vectorised over box pairs in numpy float32.
# --------------------------------------------------------------------------------------
def _rot_corners(b):
    """(n,5) xyxyr float32 -> (n,4,2) corners rotated about the centre with the kernel's rotate_around_center."""
    f = np.float32
    cx, cy = (b[:, 0] + b[:, 2]) / f(2), (b[:, 1] + b[:, 3]) / f(2)
    xs = np.stack([b[:, 0], b[:, 2], b[:, 2], b[:, 0]], 1)
    ys = np.stack([b[:, 1], b[:, 1], b[:, 3], b[:, 3]], 1)
    cs, sn = np.cos(b[:, 4]).astype(f)[:, None], np.sin(b[:, 4]).astype(f)[:, None]
    dx, dy = xs - cx[:, None], ys - cy[:, None]
    return np.stack([dx * cs + dy * sn + cx[:, None], -dx * sn + dy * cs + cy[:, None]], -1).astype(f)


def _in_box(box, pts):
    """box (n,1,5) xyxyr, pts (n|1, m, 4, 2) -> bool: the kernel's check_in_box2d (rotate the point by -angle, 1e-5 margin)."""
    f = np.float32
    cx, cy = (box[..., 0] + box[..., 2]) / f(2), (box[..., 1] + box[..., 3]) / f(2)
    cs, sn = np.cos(-box[..., 4]).astype(f), np.sin(-box[..., 4]).astype(f)
    dx, dy = pts[..., 0] - cx[..., None], pts[..., 1] - cy[..., None]
    rx = dx * cs[..., None] + dy * sn[..., None] + cx[..., None]
    ry = -dx * sn[..., None] + dy * cs[..., None] + cy[..., None]
    m = f(1e-5)
    return ((rx > box[..., 0, None] - m) & (rx < box[..., 2, None] + m) & (ry > box[..., 1, None] - m)
            & (ry < box[..., 3, None] + m))


def boxes_iou_bev(a, b):
    """mmdet3d `boxes_iou_bev(boxes_a (N,5), boxes_b (M,5))` -> (N,M) float32 rotated BEV IoU (iou3d_kernel.cu box_overlap)."""
    f = np.float32
    a, b = np.asarray(a, dtype=f).reshape(-1, 5), np.asarray(b, dtype=f).reshape(-1, 5)
    overlap = boxes_overlap_bev(a, b)
    sa = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None]
    sb = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))[None, :]
    return (overlap / np.maximum(sa + sb - overlap, f(1e-8))).astype(f)


def boxes_overlap_bev(a, b):
    """mmdet3d `boxes_overlap_bev_gpu`: rotated BEV overlap AREA of (N,5) x (M,5) xyxyr boxes (iou3d_kernel.cu box_overlap)."""
    f = np.float32
    a, b = np.asarray(a, dtype=f).reshape(-1, 5), np.asarray(b, dtype=f).reshape(-1, 5)
    N, M = len(a), len(b)
    if N == 0 or M == 0:
        return np.zeros((N, M), f)
    A, B = _rot_corners(a), _rot_corners(b)
    A5, B5 = np.concatenate([A, A[:, :1]], 1), np.concatenate([B, B[:, :1]], 1)
    # edge i of a: p0 = A[i], p1 = A[i+1]; edge j of b: q0 = B[j], q1 = B[j+1]   -> (N, M, 4, 4, 2)
    p0, p1 = A5[:, None, :4, None, :], A5[:, None, 1:, None, :]
    q0, q1 = B5[None, :, None, :4, :], B5[None, :, None, 1:, :]

    def cross3(u, v, o):
        return (u[..., 0] - o[..., 0]) * (v[..., 1] - o[..., 1]) - (v[..., 0] - o[..., 0]) * (u[..., 1] - o[..., 1])
    rect = ((np.minimum(p0[..., 0], p1[..., 0]) <= np.maximum(q0[..., 0], q1[..., 0]))
            & (np.minimum(q0[..., 0], q1[..., 0]) <= np.maximum(p0[..., 0], p1[..., 0]))
            & (np.minimum(p0[..., 1], p1[..., 1]) <= np.maximum(q0[..., 1], q1[..., 1]))
            & (np.minimum(q0[..., 1], q1[..., 1]) <= np.maximum(p0[..., 1], p1[..., 1])))
    s1, s2, s3, s4 = cross3(q0, p1, p0), cross3(p1, q1, p0), cross3(p0, q1, q0), cross3(q1, p1, q0)
    hit = rect & (s1 * s2 > 0) & (s3 * s4 > 0)
    s5 = cross3(q1, p1, p0)
    with np.errstate(divide='ignore', invalid='ignore'):
        den = s5 - s1
        ix = (s5 * q0[..., 0] - s1 * q1[..., 0]) / den
        iy = (s5 * q0[..., 1] - s1 * q1[..., 1]) / den
        a0, b0 = p0[..., 1] - p1[..., 1], p1[..., 0] - p0[..., 0]
        c0 = p0[..., 0] * p1[..., 1] - p1[..., 0] * p0[..., 1]
        a1, b1 = q0[..., 1] - q1[..., 1], q1[..., 0] - q0[..., 0]
        c1 = q0[..., 0] * q1[..., 1] - q1[..., 0] * q0[..., 1]
        D = a0 * b1 - a1 * b0
        small = np.abs(den) <= f(1e-8)
        ix = np.where(small, (b0 * c1 - b1 * c0) / D, ix)
        iy = np.where(small, (a1 * c0 - a0 * c1) / D, iy)
    inter = np.stack([ix, iy], -1).reshape(N, M, 16, 2)
    b_in_a = _in_box(a[:, None, :], np.broadcast_to(B[None], (N, M, 4, 2)))        # corners of b inside a
    a_in_b = _in_box(np.broadcast_to(b[None, :, :], (N, M, 5)), np.broadcast_to(A[:, None], (N, M, 4, 2)))
    pts = np.concatenate([inter, np.broadcast_to(B[None], (N, M, 4, 2)), np.broadcast_to(A[:, None], (N, M, 4, 2))], 2)
    valid = np.concatenate([hit.reshape(N, M, 16), b_in_a, a_in_b], 2)             # (N, M, 24)
    pts = np.where(valid[..., None], pts, f(0)).astype(f)
    cnt = valid.sum(-1)
    centre = pts.sum(2) / np.maximum(cnt, 1)[..., None].astype(f)
    ang = np.arctan2(pts[..., 1] - centre[..., None, 1], pts[..., 0] - centre[..., None, 0]).astype(f)
    ang = np.where(valid, ang, f(10))                                              # invalid points sort last
    order = np.argsort(ang, axis=2, kind='stable')
    pts = np.take_along_axis(pts, order[..., None], 2)
    vs = np.take_along_axis(valid, order, 2)
    d = pts - pts[:, :, :1]
    tri = d[:, :, :-1, 0] * d[:, :, 1:, 1] - d[:, :, :-1, 1] * d[:, :, 1:, 0]
    tri = np.where(vs[:, :, :-1] & vs[:, :, 1:], tri, f(0))
    return (np.abs(tri.sum(-1, dtype=f)) / f(2)).astype(f)


def nms_bev(boxes, scores, thresh, pre_maxsize=None, post_max_size=None, iou=None):
    """mmdet3d `nms_gpu(boxes (n,5) xyxyr, scores, thresh, pre_maxsize, post_max_size)` -> kept original indices, best first."""
    order = np.argsort(-np.asarray(scores), kind='stable')
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    bx = np.asarray(boxes, dtype=np.float32)[order]
    iou = boxes_iou_bev(bx, bx) if iou is None else iou[np.ix_(order, order)]
    removed = np.zeros(len(order), dtype=bool)
    keep = []
    for i in range(len(order)):
        if removed[i]:
            continue
        keep.append(int(order[i]))
        removed[i + 1:] |= iou[i, i + 1:] > np.float32(thresh)
    return keep if post_max_size is None else keep[:post_max_size]


def xywhr2xyxyr(bev):
    out = bev.clone()
    hw, hl = bev[:, 2] / 2, bev[:, 3] / 2
    out[:, 0], out[:, 1], out[:, 2], out[:, 3] = bev[:, 0] - hw, bev[:, 1] - hl, bev[:, 0] + hw, bev[:, 1] + hl
    return out


def get_bboxes_rotate_nms(dicts, cfg, pre_maxsize, post_maxsize):
    """FD:1347-1393 with nms_type='rotate': per task nms_gpu on xywhr2xyxyr(boxes.bev), thresh = the task's radius."""
    res = []
    for d in dicts:
        b, s, l = d['bboxes'], d['scores'], d['labels']
        keep_mask = torch.zeros_like(s, dtype=torch.bool)
        for idx, radius in NMS_TASKS[cfg.dataset]:
            task_mask = torch.zeros_like(s, dtype=torch.bool)
            for c in idx:
                task_mask |= l == c
            if radius > 0:
                bev = xywhr2xyxyr(b[task_mask][:, [0, 1, 3, 4, 6]])
                kept = torch.tensor(nms_bev(bev.numpy(), s[task_mask].numpy(), radius, pre_maxsize, post_maxsize), dtype=torch.long)
            else:
                kept = torch.arange(int(task_mask.sum()))
            if kept.numel():
                keep_mask[torch.where(task_mask)[0][kept]] = True
        b, s, l = b[keep_mask], s[keep_mask], l[keep_mask]
        if len(b) > 200:
            inds = s.argsort(descending=True)[:200]
            b, s, l = b[inds], s[inds], l[inds]
        res.append((b, s, l.int()))
    return res


def bbox3d_mapping_back(boxes, scale_factor, flip_horizontal, flip_vertical):
    """mmdet3d 0.17.1 `bbox3d_mapping_back` on a LiDAR box tensor (flip: cols 1::7 / 0::7 negated, yaw -> -yaw (+ pi))."""
    b = boxes.clone()
    if flip_horizontal:
        b[:, 1::7] = -b[:, 1::7]
        b[:, 6] = -b[:, 6] + np.pi
    if flip_vertical:
        b[:, 0::7] = -b[:, 0::7]
        b[:, 6] = -b[:, 6]
    b[:, :6] *= 1 / scale_factor
    b[:, 7:] *= 1 / scale_factor
    return b


def merge_aug_boxes(aug_boxes, aug_scores, aug_labels):
    """merge_augs.py:113-184 on the concatenated mapped-back detections (constants hard-wired there: rotate NMS at 0.1,
    voting at IoU >= 0.65 without score voting, max_num 500)."""
    if len(aug_labels) == 0:
        return aug_boxes, aug_scores, aug_labels
    for_nms = xywhr2xyxyr(aug_boxes[:, [0, 1, 3, 4, 6]])
    mb, ms, ml = [], [], []
    for cls in range(int(aug_labels.max()) + 1):
        sel = aug_labels == cls
        if not sel.any():
            continue
        boxes_i, nms_i, scores_i, labels_i = aug_boxes[sel], for_nms[sel], aug_scores[sel], aug_labels[sel]
        selected = torch.tensor(nms_bev(nms_i.numpy(), scores_i.numpy(), 0.1), dtype=torch.long)
        chosen = boxes_i[selected]
        iou = torch.from_numpy(boxes_iou_bev(xywhr2xyxyr(chosen[:, [0, 1, 3, 4, 6]]).numpy(), nms_i.numpy()))
        iou[iou < 0.65] = 0.
        voted = (iou[:, :, None] * boxes_i[None]).sum(dim=1) / (iou[:, :, None].sum(dim=1) + 1e-6)
        voted[:, 6] = torch.atan2((iou * torch.sin(boxes_i[None, :, 6])).sum(dim=1) / (iou.sum(dim=1) + 1e-6),
                                  (iou * torch.cos(boxes_i[None, :, 6])).sum(dim=1) / (iou.sum(dim=1) + 1e-6))
        mb.append(voted)
        ms.append(scores_i[selected])
        ml.append(labels_i[selected])
    mb, ms, ml = torch.cat(mb), torch.cat(ms), torch.cat(ml)
    order = ms.sort(0, descending=True)[1][:min(500, len(aug_boxes))]
    return mb[order], ms[order], ml[order]


# --------------------------------------------------------------------------------------
# FocalEncoder neck (projects/mmdet3d_plugin/models/necks/focal_encoder.py:15-222), inference.
# The two torchvision blocks it instantiates (un-vendored third party; torchvision is not installed here) are
# restated from their published definitions: mobilenetv2.InvertedResidual and resnet.BasicBlock.
# --------------------------------------------------------------------------------------
def _bn2d(x, sd, p):
    return F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'], sd.get(p + 'weight'), sd.get(p + 'bias'),
                        False, 0.0, BN_EPS)


def inverted_residual(x, sd, p, inp, oup, expand_ratio):
    """torchvision InvertedResidual (stride 1): [1x1 expand + BN + ReLU6] -> 3x3 depthwise + BN + ReLU6 -> 1x1 + BN."""
    y, i = x, 0
    if expand_ratio != 1:
        y = F.relu6(_bn2d(F.conv2d(y, sd[f'{p}conv.0.0.weight']), sd, f'{p}conv.0.1.'))
        i = 1
    w = sd[f'{p}conv.{i}.0.weight']
    y = F.relu6(_bn2d(F.conv2d(y, w, padding=1, groups=w.shape[0]), sd, f'{p}conv.{i}.1.'))
    y = _bn2d(F.conv2d(y, sd[f'{p}conv.{i + 1}.weight']), sd, f'{p}conv.{i + 2}.')
    return x + y if inp == oup else y


def basic_block(x, sd, p):
    """torchvision resnet.BasicBlock without downsample."""
    y = F.relu(_bn2d(F.conv2d(x, sd[p + 'conv1.weight'], padding=1), sd, p + 'bn1.'))
    y = _bn2d(F.conv2d(y, sd[p + 'conv2.weight'], padding=1), sd, p + 'bn2.')
    return F.relu(y + x)


def conv_bn(x, sd, p, k):
    """EU:10-33 ConvBNReLU with activation_layer=None."""
    return _bn2d(F.conv2d(x, sd[p + 'conv.weight'], sd.get(p + 'conv.bias'), padding=k // 2), sd, p + 'bn.')


def focal_encoder_forward(sd, cfg, img_feats, pts_feats, lidar2img=None, input_shape=None, img_aug=None, taps=None):
    """focal_encoder.py:171-222 (+ FocalEncoderLayer.forward :52-87) for input_pts=True.
    cfg: dict(num_layers, hidden_channel, iterbev, max_points_height, multistage_heatmap, input_img, iterbev_wo_img,
    extra_feat, iter_bev_cam[, cam_lss, pc_range, img_scale]).  Returns (new_img_feat, [pts_feat_conv, stage maps | tensor]).
    With cam_lss (requires iter_bev_cam, as in FocalFormer3D_LC.py) the image branch is the Lift-Splat-Shoot BEV map built
    from the inverse lidar2img matrices (focal_encoder.py:175-193)."""
    C = cfg['hidden_channel']
    if cfg['input_img'] and cfg.get('cam_lss'):
        B = pts_feats.shape[0]
        inv = torch.inverse(lidar2img.float())
        lcfg = dict(img_scale=cfg['img_scale'], downsample=4, depth_range=[4.0, 45.0, 1.0], pc_range=cfg['pc_range'],
                    grid=0.6, camC=64)
        new_img, _ = lss_forward(sd, lcfg, img_feats.view(B, -1, *img_feats.shape[-3:]), inv[..., :3, :3].contiguous(),
                                 inv[..., :3, 3].contiguous(), img_aug, p='cam_lss.')
    else:
        new_img = F.conv2d(img_feats, sd['shared_conv_img.weight'], sd['shared_conv_img.bias'], padding=1) \
            if cfg['input_img'] else None
    new_pts = F.conv2d(pts_feats, sd['shared_conv_pts.weight'], sd['shared_conv_pts.bias'], padding=1)
    pts_feat_conv = new_pts.clone()
    if not (cfg['input_img'] or cfg['iterbev_wo_img']):
        return None, [new_pts, None]
    B = new_pts.shape[0]
    stages = []
    for i in range(cfg['num_layers']):
        p = f'fusion_blocks.{i}.'
        lidar = new_pts
        if not cfg['iterbev_wo_img']:
            if cfg['iter_bev_cam'] and (i > 0 or cfg.get('cam_lss')):
                i2p_feat = new_img
            else:
                img5 = new_img.view(B, -1, *new_img.shape[1:])
                i2p_feat = i2p_forward(sd, lidar, img5, lidar2img, input_shape, cfg['max_points_height'], img_aug,
                                       p=p + 'I2P_block.learnedAlign.')
                if taps is not None:
                    taps[f'i2p/{i}'] = i2p_feat
                if cfg['iter_bev_cam']:
                    new_img = i2p_feat
        else:
            i2p_feat = lidar
        if cfg['iterbev'] == 'bevfusion':
            sub = {k[len(p + 'P_IML.'):]: v for k, v in sd.items() if k.startswith(p + 'P_IML.')}
            p2p = local_context_attention(sub, lidar, lidar, 9)
            aug = conv_bn(torch.cat((i2p_feat, p2p), 1), sd, p + 'P_out_proj.', 1)
            new_pts = conv_bn(torch.cat((aug, lidar), 1), sd, p + 'P_integration.', 1)
        else:  # 'bevfusionmb2'
            p2p = inverted_residual(lidar, sd, p + 'P_IML.', C, C, 2)
            aug = inverted_residual(torch.cat((i2p_feat, p2p), 1), sd, p + 'P_out_proj.', 2 * C, C, 1)
            new_pts = inverted_residual(torch.cat((aug, lidar), 1), sd, p + 'P_integration.', 2 * C, C, 1)
        if not cfg['iterbev_wo_img']:
            new_img = basic_block(new_img, sd, p + 'iterimg_conv.0.')
        if cfg['multistage_heatmap']:
            stages.append(new_pts)
    if cfg['multistage_heatmap']:
        if cfg['extra_feat']:
            stages.append(conv_bn(stages[-1], sd, 'extra_output.', 3))
        return new_img, [pts_feat_conv, stages]
    return new_img, [pts_feat_conv, new_pts]


# --------------------------------------------------------------------------------------
# Lift-Splat-Shoot camera branch (projects/mmdet3d_plugin/models/necks/lss.py:125-383), inference, no point-cloud
# augmentation (apply_3d_transformation = identity at test time, A.5).
# --------------------------------------------------------------------------------------
def lss_grid(pc_range, grid):
    """lss.py:82-87 gen_dx_bx: dx, bx (cell centres of the first cell), nx (LongTensor truncation)."""
    bounds = [[pc_range[0], pc_range[3], grid], [pc_range[1], pc_range[4], grid], [pc_range[2], pc_range[5], grid]]
    dx = torch.tensor([r[2] for r in bounds], dtype=torch.float32)
    bx = torch.tensor([r[0] + r[2] / 2.0 for r in bounds], dtype=torch.float32)
    nx = torch.tensor([int((r[1] - r[0]) / r[2]) for r in bounds], dtype=torch.long)
    return dx, bx, nx


def lss_frustum(img_scale, downsample, depth_range):
    """lss.py:217-230: (D, fH, fW, 3) image-plane points (x_px, y_px, depth)."""
    ogfH, ogfW = img_scale
    fH, fW = ogfH // downsample, ogfW // downsample
    ds = torch.arange(*depth_range, dtype=torch.float).view(-1, 1, 1).expand(-1, fH, fW)
    D = ds.shape[0]
    xs = torch.linspace(0, ogfW - 1, fW, dtype=torch.float).view(1, 1, fW).expand(D, fH, fW)
    ys = torch.linspace(0, ogfH - 1, fH, dtype=torch.float).view(1, fH, 1).expand(D, fH, fW)
    return torch.stack((xs, ys, ds), -1)


def lss_geometry(frustum, rots, trans, img_aug=None, img_metas=None):
    """lss.py:232-276 get_geometry -> (B, N, D, fH, fW, 3) ego-frame points."""
    B, N, _ = trans.shape
    if img_aug is not None:
        post_rots, post_trans = img_aug[..., :3, :3], img_aug[..., :3, 3]
        pts = frustum - post_trans.view(B, N, 1, 1, 1, 3)
        pts = torch.inverse(post_rots).view(B, N, 1, 1, 1, 3, 3).matmul(pts.unsqueeze(-1))
    else:
        pts = frustum.repeat(B, N, 1, 1, 1, 1).unsqueeze(-1)
    pts = torch.cat((pts[..., :2, :] * pts[..., 2:3, :], pts[..., 2:3, :]), 5)
    pts = rots.view(B, N, 1, 1, 1, 3, 3).matmul(pts).squeeze(-1)
    pts = pts + trans.view(B, N, 1, 1, 1, 3)
    if img_metas is not None:                                   # lss.py:262-265: into the augmented LiDAR frame
        pts = torch.stack([apply_3d_transformation(pts[b].reshape(-1, 3), img_metas[b], reverse=False).view(pts.shape[1:])
                           for b in range(B)])
    return pts


def lss_forward(sd, cfg, x, rots, trans, img_aug=None, p='', img_metas=None, taps=None):
    """lss.py:377-383 LiftSplatShoot.forward.  x (B, N, inputC, fH, fW); cfg: dict(img_scale, downsample, depth_range,
    pc_range, grid, camC).  Voxel pooling as exact per-cell sums (lss.py:324-362 computes the same sums with a cumsum
    trick whose fp32 cancellation noise is not reproduced).  Returns (bev (B, outC, X, Y), depth (B, N, D, fH, fW))."""
    B, N, Cin, fH, fW = x.shape
    camC = cfg['camC']
    frustum = lss_frustum(cfg['img_scale'], cfg['downsample'], cfg['depth_range'])
    D = frustum.shape[0]
    dx, bx, nx = lss_grid(cfg['pc_range'], cfg['grid'])
    geom = lss_geometry(frustum, rots, trans, img_aug, img_metas)
    y = F.conv2d(x.view(B * N, Cin, fH, fW), sd[p + 'camencode.depthnet.weight'], sd[p + 'camencode.depthnet.bias'])
    depth = y[:, :D].softmax(dim=1)                                          # lss.py:132-141
    feat = depth.unsqueeze(1) * y[:, D:D + camC].unsqueeze(2)               # (BN, camC, D, fH, fW)
    feat = feat.view(B, N, camC, D, fH, fW).permute(0, 1, 3, 4, 5, 2).reshape(-1, camC)
    cell = ((geom - (bx - dx / 2.0)) / dx).long().view(-1, 3)
    batch_ix = torch.arange(B).repeat_interleave(N * D * fH * fW)
    kept = ((cell[:, 0] >= 0) & (cell[:, 0] < nx[0]) & (cell[:, 1] >= 0) & (cell[:, 1] < nx[1])
            & (cell[:, 2] >= 0) & (cell[:, 2] < nx[2]))
    X, Y, Z = int(nx[0]), int(nx[1]), int(nx[2])
    flat = ((batch_ix[kept] * Z + cell[kept, 2]) * X + cell[kept, 0]) * Y + cell[kept, 1]
    vox = torch.zeros(B * Z * X * Y, camC, dtype=torch.float64)
    vox.index_add_(0, flat, feat[kept].double())
    vox = vox.to(x.dtype).view(B, Z, X, Y, camC).permute(0, 4, 1, 2, 3)      # (B, C, Z, X, Y), lss.py:358-360
    if taps is not None:
        taps['vox'] = vox
    bev = vox.reshape(B, camC * Z, X, Y).permute(0, 1, 3, 2)                 # s2c, lss.py:371-375
    q = p + 'bevencode.'
    for i in range(0, 12, 3):
        bev = F.relu(_bn2d(F.conv2d(bev, sd[f'{q}{i}.weight'], padding=1), sd, f'{q}{i + 1}.'))
    return bev, depth.view(B, N, D, fH, fW)
