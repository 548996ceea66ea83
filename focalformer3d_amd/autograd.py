"""Differentiable multi-scale deformable attention on MI355X - first brick of the training path.

``MultiScaleDeformableAttnFunction`` mirrors mmcv 1.3.18 ``mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttnFunction``
(un-vendored; Appendix A.3): same ``apply(value, value_spatial_shapes, value_level_start_index, sampling_locations,
attention_weights, im2col_step)`` signature, forward = ``ff3d_msda_fwd``, backward = ``ff3d_msda_bwd`` (the counterparts of
``ext_module.ms_deform_attn_forward / _backward``).

``RoIGridSampleFunction``: the RoI feature read of FD:890-919 (box decode, g x g grid, bilinear sampling of every pyramid level)
as one differentiable op on the channels-last pyramid: forward = ``ff3d_roi_grid_sample``, backward =
``ff3d_roi_grid_sample_bwd`` (the reference differentiates through ``F.grid_sample``; the boxes are detached, FD:956).

``LinearWgradFunction`` / ``train_linear`` (round 6): ``F.linear`` whose WEIGHT and BIAS gradients run on the own split-fp16
"TN" kernel (``ff3d_linear_wgrad_f16x3``, csrc/wgrad.hip) instead of the framework's fp32 GEMM + column-sum reduce; forward and
input gradient stay the framework's GEMMs.  Used where the row count makes the vendor's choice slow: the per-layer ``value_proj``
over the flattened BEV pyramid (FD:927-933 under autograd: M = frames x 42 525 rows).

The training-mode forward that uses them is focalformer3d_amd/train_forward.py; the inference modules do not route through
autograd.
"""
import os
import weakref

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops


# rows from which train_linear takes the own weight-gradient kernel (below: the framework's GEMM wins on launch count; measured
# on the decoder's shapes, profiles/r06_train_step_kernels.txt); FF3D_WGRAD_MIN_ROWS=0 disables the own kernel
WGRAD_MIN_ROWS = int(os.environ.get('FF3D_WGRAD_MIN_ROWS', '16384'))
_X_AMAX = [None]            # (weakref to the last input measured, its version, its partial maxima)


def _input_amax(x, x2):
    """rows_absmax of a layer input, measured once per tensor OBJECT and version: the layers of the decoder that read the same
    flattened pyramid share one pass (a new training step builds a new tensor, so nothing stale can be reused)."""
    memo = _X_AMAX[0]
    if memo is not None and memo[0]() is x and memo[1] == x._version and memo[2].device == x.device:
        return memo[2]
    amax = ops.rows_absmax(x2)
    _X_AMAX[0] = (weakref.ref(x), x._version, amax)
    return amax


class LinearWgradFunction(Function):
    """``y = x W^T + b``; backward: dx = dy W (the framework's GEMM), (dW, db) = ff3d_linear_wgrad_f16x3(x, dy)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1])
        ctx.save_for_backward(x, weight, _input_amax(x, x2))
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, amax_x = ctx.saved_tensors
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = dy.matmul(weight)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            x2, dy2 = x.reshape(-1, x.shape[-1]), dy.reshape(-1, dy.shape[-1])
            if not ops.linear_wgrad_ok(x2, dy2):
                x2, dy2 = x2.contiguous(), dy2.contiguous()
            dw, db = ops.linear_wgrad(x2, dy2, want_bias=ctx.has_bias and ctx.needs_input_grad[2], amax_x=amax_x)
            if not ctx.needs_input_grad[1]:
                dw = None
        return dx, dw, db


def train_linear(x, weight, bias=None):
    """``F.linear`` for the training path: the own weight-gradient kernel from WGRAD_MIN_ROWS rows, the framework's op otherwise."""
    rows = x.numel() // max(x.shape[-1], 1)
    if (WGRAD_MIN_ROWS > 0 and rows >= WGRAD_MIN_ROWS and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and torch.is_grad_enabled() and (weight.requires_grad or (bias is not None and bias.requires_grad))
            and x.shape[-1] % 4 == 0 and weight.shape[0] % 4 == 0 and x.shape[-1] >= 4 and weight.shape[0] >= 4):
        return LinearWgradFunction.apply(x, weight, bias)
    return F.linear(x, weight, bias)


def _level_hw(spatial_shapes):
    if isinstance(spatial_shapes, torch.Tensor):
        return [tuple(int(v) for v in r) for r in spatial_shapes.tolist()]
    return [tuple(int(v) for v in r) for r in spatial_shapes]


class MultiScaleDeformableAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step=64):
        """value (B, Nv, heads, Dh), sampling_locations (B, Nq, heads, L, P, 2) in [0, 1], attention_weights
        (B, Nq, heads, L, P) -> (B, Nq, heads*Dh).  ``value_level_start_index`` / ``im2col_step`` are accepted for API parity
        (the level offsets follow from the shapes; the kernels do not tile the batch)."""
        value, loc, w = value.contiguous(), sampling_locations.contiguous(), attention_weights.contiguous()
        ctx.save_for_backward(value, loc, w)
        if isinstance(value_spatial_shapes, torch.Tensor) and value_spatial_shapes.is_cuda:
            # mmcv's calling convention (device int64 tables, FD:837-841): the kernel reads them in place - no .tolist(),
            # no host sync, legal under graph capture.  Only backward (training, never captured) needs the host copy.
            ctx.level_hw, ctx.shapes = None, value_spatial_shapes
            if value_level_start_index is None:
                hw = value_spatial_shapes[:, 0] * value_spatial_shapes[:, 1]
                value_level_start_index = torch.cat([hw.new_zeros(1), hw.cumsum(0)[:-1]])
            return ops.msda_fwd_dev(value, value_spatial_shapes.contiguous(), value_level_start_index.contiguous(), loc, w)
        ctx.level_hw = _level_hw(value_spatial_shapes)
        return ops.msda_fwd(value, ctx.level_hw, loc, w)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, loc, w = ctx.saved_tensors
        if ctx.level_hw is None:
            ctx.level_hw = _level_hw(ctx.shapes)
        gv, gl, gw = ops.msda_bwd(value, ctx.level_hw, loc, w, grad_output.contiguous())
        return gv, None, None, gl, gw, None


class RoIGridSampleFunction(Function):
    @staticmethod
    def forward(ctx, feat_cl, query_box, level_hw, g, expand, coder, roi_range, layout=1):
        """feat_cl (B, Nv, C) channels-last pyramid, query_box (B, >=8, Nq) raw (detached) predictions ->
        (B*Nq, L*C*g*g) RoI matrix, columns [level][point][channel] (layout 1) or [level][channel][point] (0, FD:919)."""
        feat_cl, query_box = feat_cl.contiguous(), query_box.detach().contiguous()
        ctx.save_for_backward(query_box)
        ctx.meta = (tuple(feat_cl.shape), [tuple(hw) for hw in level_hw], g, expand, tuple(coder), tuple(roi_range), layout)
        return ops.roi_grid_sample(feat_cl, ctx.meta[1], query_box, g, expand, coder, roi_range, layout=layout)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (query_box,) = ctx.saved_tensors
        shape, level_hw, g, expand, coder, roi_range, layout = ctx.meta
        grad = ops.roi_grid_sample_bwd(grad_output.contiguous(), shape, level_hw, query_box, g, expand, coder, roi_range, layout)
        return grad, None, None, None, None, None, None, None


def _dropout_keep(shape, p, device):
    """uint8 keep-mask of an attention dropout with drop probability ``p``, drawn with the framework's generator straight into
    uint8 (no fp32 intermediate of the same shape).  A module-level function so that tests can pin the draw."""
    return torch.empty(shape, dtype=torch.uint8, device=device).bernoulli_(1.0 - p)


class MaskedSelfAttentionFunction(Function):
    """softmax(q k^T / sqrt(Dh) + mask) v per head with attention dropout, forward and backward on libff3d_hip.so
    (ff3d_mha_train_fwd / _bwd): the scaled-dot-product core of ``nn.MultiheadAttention`` on the training route.
    q, k, v (B, N, C); mask (B, N, N) bool / uint8 (True = blocked) or None; the dropout keep-mask is drawn with the framework's
    generator (``Tensor.bernoulli_`` straight into uint8 - no (B, heads, N, N) fp32 intermediate: 128 MB at 4 frames x 1000
    queries), so seeding behaves as for any other dropout."""

    @staticmethod
    def forward(ctx, q, k, v, heads, mask, dropout_p):
        mask8 = None if mask is None else mask.to(torch.uint8).contiguous()
        keep, keep_scale = None, 1.0
        if dropout_p > 0.0:
            B, N, _ = q.shape
            keep = _dropout_keep((B, heads, N, N), dropout_p, q.device)
            keep_scale = 1.0 / (1.0 - dropout_p)
        out, lse = ops.mha_train_fwd(q, k, v, heads, mask8, keep, keep_scale)
        ctx.save_for_backward(q, k, v, out, lse, mask8, keep)
        ctx.heads, ctx.keep_scale = heads, keep_scale
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        q, k, v, out, lse, mask8, keep = ctx.saved_tensors
        gq, gk, gv = ops.mha_train_bwd(q, k, v, ctx.heads, out, lse, grad_out.contiguous(), mask8, keep, ctx.keep_scale)
        return gq, gk, gv, None, None, None


class SimilarFunction(Function):
    """``similarFunction`` (encoder_utils.py:61-83) over libff3d_hip.so: forward = locatt_ops similar_forward (cc2k), backward =
    similar_backward(is_ori=True / False) = ck2c_ori / ck2c_loc (ff3d.h)."""

    @staticmethod
    def forward(ctx, x_ori, x_loc, kH, kW):
        x_ori, x_loc = x_ori.contiguous(), x_loc.contiguous()
        ctx.save_for_backward(x_ori, x_loc)
        ctx.kHW = (kH, kW)
        return ops.locatt_similar(x_ori, x_loc, kH, kW)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_outputs):
        x_ori, x_loc = ctx.saved_tensors
        kH, kW = ctx.kHW
        g = grad_outputs.contiguous()
        return ops.locatt_weighting(x_loc, g, kH, kW), ops.locatt_ck2c_loc(x_ori, g, kH, kW), None, None


class WeightingFunction(Function):
    """``weightingFunction`` (encoder_utils.py:86-106): forward = weighting_forward (ck2c_ori), backward =
    weighting_backward_ori (ck2c_loc) and weighting_backward_weight (cc2k)."""

    @staticmethod
    def forward(ctx, x_ori, x_weight, kH, kW):
        x_ori, x_weight = x_ori.contiguous(), x_weight.contiguous()
        ctx.save_for_backward(x_ori, x_weight)
        ctx.kHW = (kH, kW)
        return ops.locatt_weighting(x_ori, x_weight, kH, kW)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_outputs):
        x_ori, x_weight = ctx.saved_tensors
        kH, kW = ctx.kHW
        g = grad_outputs.contiguous()
        return ops.locatt_ck2c_loc(g, x_weight, kH, kW), ops.locatt_similar(g, x_ori, kH, kW), None, None


class BevPoolFunction(Function):
    """``QuickCumsumCuda`` (ops/bev_pool/bev_pool_op.py:37-88): x (n, c) sorted by rank, geom_feats (n, 4), ranks (n) ->
    (B, D, H, W, c); backward = ff3d_bev_pool_bwd."""

    @staticmethod
    def forward(ctx, x, geom_feats, ranks, B, D, H, W):
        kept = torch.ones(x.shape[0], device=x.device, dtype=torch.bool)
        kept[1:] = ranks[1:] != ranks[:-1]
        interval_starts = torch.where(kept)[0].int()
        interval_lengths = torch.zeros_like(interval_starts)
        interval_lengths[:-1] = interval_starts[1:] - interval_starts[:-1]
        interval_lengths[-1] = x.shape[0] - interval_starts[-1]
        geom_feats = geom_feats.int().contiguous()
        ctx.save_for_backward(interval_starts, interval_lengths, geom_feats)
        ctx.saved_shapes = B, D, H, W
        return ops.bev_pool_forward(x.contiguous(), geom_feats, interval_lengths, interval_starts, B, D, H, W)

    @staticmethod
    @once_differentiable
    def backward(ctx, out_grad):
        interval_starts, interval_lengths, geom_feats = ctx.saved_tensors
        B, D, H, W = ctx.saved_shapes
        x_grad = ops.bev_pool_backward(out_grad.contiguous(), geom_feats, interval_lengths, interval_starts, B, D, H, W)
        return x_grad, None, None, None, None, None, None


def bev_pool(feats, coords, B, D, H, W):
    """Differentiable ``bev_pool`` (ops/bev_pool/bev_pool_op.py:91-110): rank, sort, QuickCumsumCuda, (B, c, D, H, W)."""
    assert feats.shape[0] == coords.shape[0]
    ranks = coords[:, 0] * (W * D * B) + coords[:, 1] * (D * B) + coords[:, 2] * B + coords[:, 3]
    indices = ranks.argsort()
    feats, coords, ranks = feats[indices], coords[indices], ranks[indices]
    x = BevPoolFunction.apply(feats, coords, ranks, B, D, H, W)
    return x.permute(0, 4, 1, 2, 3).contiguous()
