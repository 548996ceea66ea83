"""``LocalContextAttentionBlock`` and the two ``locatt_ops`` operators on MI355X.

Mirror of projects/mmdet3d_plugin/models/utils/encoder_utils.py:10-33 (ConvBNReLU), :61-106 (similarFunction /
weightingFunction over the CUDA extension ops/locatt_ops) and :109-163 (LocalContextAttentionBlock) - same constructor
arguments and parameter names - used by the `iterbev='bevfusion'` fusion blocks of the FocalEncoder neck
(SURVEY.md §8f rank 1).  ``.eval()``: BatchNorm folded, one fused attention kernel.  ``.train()``: the reference's op sequence
under autograd - ``similarFunction`` / ``weightingFunction`` on the HIP forward AND backward kernels (autograd.py), BatchNorm on
batch statistics, softmax by the framework.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .layers import weight_signature


DENSE_MODE = os.environ.get('FF3D_DENSE_MODE', 'f16x3')     # neck 3x3 convs: 'f16x3' (own MFMA kernels) | 'vendor'


class ConvBNReLU(nn.Module):
    """encoder_utils.py:10-33."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, groups=1,
                 norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU, bias='auto', inplace=True, affine=True):
        super().__init__()
        padding = dilation * (kernel_size - 1) // 2
        self.use_norm = norm_layer is not None
        self.use_activation = activation_layer is not None
        if bias == 'auto':
            bias = not self.use_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation=dilation, groups=groups,
                              bias=bias)
        if self.use_norm:
            self.bn = norm_layer(out_channels, affine=affine)
        if self.use_activation:
            self.activation = activation_layer(inplace=inplace)

    def folded(self):
        w, b = self.conv.weight, self.conv.bias
        if not self.use_norm:
            return w, b
        bn = self.bn
        keep = not torch.is_grad_enabled()             # kept per weight version (layers.weight_signature), see focal_encoder._fold
        if keep:
            src = [t for t in (w, b, bn.weight, bn.bias, bn.running_mean, bn.running_var) if t is not None]
            sig = (bn.eps,) + weight_signature(src)
            hit = self.__dict__.get('_ff3d_fold')
            if hit is not None and hit[0] == sig:
                return hit[1], hit[2]
        g = bn.weight if bn.weight is not None else torch.ones_like(bn.running_var)
        beta = bn.bias if bn.bias is not None else torch.zeros_like(bn.running_var)
        scale = g / torch.sqrt(bn.running_var + bn.eps)
        w2 = w * scale.view(-1, 1, 1, 1)
        b2 = beta - bn.running_mean * scale if b is None else (b - bn.running_mean) * scale + beta
        w2, b2 = w2.contiguous(), b2.contiguous()
        if keep:
            self.__dict__['_ff3d_fold'] = (sig, w2, b2)
        return w2, b2

    def forward(self, x):
        """Inference form on the device: conv with the BatchNorm folded in, shift (+ ReLU) in one fused pass; dense 3x3
        convs run on the split-fp16 MFMA kernels (dense_conv3x3)."""
        if self.training:                              # EU:26-33 under autograd (batch-statistics BatchNorm)
            x = self.conv(x)
            if self.use_norm:
                x = self.bn(x)
            return self.activation(x) if self.use_activation else x
        w, b = self.folded()
        c = self.conv
        if c.kernel_size == (3, 3) and c.groups == 1 and c.dilation == (1, 1) and c.stride in ((1, 1), (2, 2)):
            return dense_conv3x3(self, x, w, b, self.use_activation, c.stride[0])
        ops.note_vendor('neck ConvBNReLU (%dx%d, groups %d)' % (*c.kernel_size, c.groups), x.shape[0] * x.shape[2] * x.shape[3], w.shape[0],
                        w.shape[1] * w.shape[2] * w.shape[3])
        y = F.conv2d(x, w, None if self.use_activation else b, c.stride, c.padding, c.dilation, c.groups)
        return ops.bias_relu_(y, b) if self.use_activation else y


def input_pair(x, hint=None):
    """NHWC (hi, lo') pair of a conv input: ``x`` itself when it already is one (a producer inside this forward handed it over), the
    pair a PRODUCER left on its NCHW output (``_ff3d_pair``, honoured while the tensor is unmodified), else one transposing split pass.
    (Nothing is cached on the tensor here: an input buffer that is refilled between calls - a captured graph's static input - must be
    converted every time.)"""
    if isinstance(x, tuple):
        return x
    p_ = getattr(x, '_ff3d_pair', None)
    if p_ is not None and p_[0].shape == (x.shape[0], x.shape[2], x.shape[3], x.shape[1]) and x._version == 0:
        return p_
    return ops.split_f16(x.contiguous(), to_nhwc=True, hint=hint)


def dense_conv3x3(owner, x, weight, bias, relu=False, stride=1, channels_last=False, pair_out=False):
    """3x3 conv (padding 1) + bias (+ ReLU) of a neck module: split-fp16 MFMA kernels (splitmm.hip / convhalo.hip,
    fp32-class) when the channel count allows and FF3D_DENSE_MODE is not 'vendor', MIOpen fp32 otherwise.  The split
    weights are cached on ``owner`` per weight version.  ``x``: an NCHW fp32 tensor or (round 5) the NHWC (hi, lo') Pair a producer
    handed over; ``pair_out``: return the result as such a Pair instead of NCHW fp32 (needs the own kernels).  ``channels_last`` (round 5): the result is still an (N, C, H, W)
    tensor for every consumer, but its MEMORY is NHWC (torch's channels_last strides, written directly by the halo kernel) -
    what the camera-projection sampler gathers from, without an NCHW tensor and a transposing pass in between."""
    if (channels_last and DENSE_MODE == 'f16x3' and weight.shape[1] % 32 == 0 and weight.shape[0] >= 64 and x.is_cuda and stride == 1
            and ops.CONV_HALO != '0'):
        key = (weight.data_ptr(), weight._version, tuple(weight.shape))
        cache = owner.__dict__.get('_split_w')
        if cache is None or cache[0] != key:
            cache = owner.__dict__['_split_w'] = (key, ops.split_weight_f16(weight, bias=bias), ops.new_hint(x.device))
        y = ops.conv3x3_f16x3(ops.split_f16(x.contiguous(), to_nhwc=True, hint=cache[2]), cache[1], bias, relu, 1, nhwc_out=True)
        return y.permute(0, 3, 1, 2)
    if DENSE_MODE == 'f16x3' and weight.shape[1] % 32 == 0 and weight.shape[0] > 16 and (x.is_cuda if torch.is_tensor(x) else True):
        key = (weight.data_ptr(), weight._version, tuple(weight.shape))
        cache = owner.__dict__.get('_split_w')
        if cache is None or cache[0] != key:
            dev = x.device if torch.is_tensor(x) else x[0].device
            cache = owner.__dict__['_split_w'] = (key, ops.split_weight_f16(weight, bias=bias), ops.new_hint(dev))
        # (cache[2]: this layer's persistent exponent guess for the input conversion - ff3d.h RANGE NORMALISATION)
        xs = input_pair(x, cache[2])
        return ops.conv3x3_f16x3(xs, cache[1], bias, relu, stride, split_out=pair_out)
    if pair_out or isinstance(x, tuple):
        raise RuntimeError('dense_conv3x3: pair input / output needs the split-fp16 kernels (channel counts % 32, dense mode f16x3)')
    ops.note_vendor('neck conv3x3', x.shape[0] * x.shape[2] * x.shape[3], weight.shape[0], 9 * weight.shape[1])
    y = F.conv2d(x, weight, None if relu else bias, stride=stride, padding=1)
    return ops.bias_relu_(y, bias) if relu else y


def similar_forward(x_ori, x_loc, kH, kW):
    """``locatt_ops.localattention.similar_forward`` (encoder_utils.py:68-69)."""
    return ops.locatt_similar(x_ori.contiguous(), x_loc.contiguous(), kH, kW)


def weighting_forward(x_ori, x_weight, kH, kW):
    """``locatt_ops.localattention.weighting_forward`` (encoder_utils.py:93-94)."""
    return ops.locatt_weighting(x_ori.contiguous(), x_weight.contiguous(), kH, kW)


class LocalContextAttentionBlock(nn.Module):
    """encoder_utils.py:109-163; forward = one fused kernel (similarity, softmax, weighting) after the 1x1 projections."""

    def __init__(self, in_channels, out_channels, kernel_size, last_affine=True, in_channels_key=None):
        super().__init__()
        if in_channels_key is None:
            in_channels_key = in_channels
        self.kernel_size = kernel_size
        cbr = lambda i, o, **kw: ConvBNReLU(i, o, kernel_size=1, norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU, **kw)
        self.query_project = nn.Sequential(cbr(in_channels, out_channels), cbr(out_channels, out_channels))
        self.key_project = nn.Sequential(cbr(in_channels_key, out_channels), cbr(out_channels, out_channels))
        self.value_project = cbr(in_channels_key, out_channels, affine=last_affine)
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if getattr(m, 'bias', None) is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, target_feats, source_feats, **kwargs):
        if not target_feats.is_cuda:
            raise RuntimeError('LocalContextAttentionBlock: inputs must live on the MI355X (HIP) device - no CPU fallback')
        if self.training:                              # EU:154-163, differentiable
            from .autograd import SimilarFunction, WeightingFunction
            query, key = self.query_project(target_feats), self.key_project(source_feats)
            value = self.value_project(source_feats)
            weight = SimilarFunction.apply(query, key, self.kernel_size, self.kernel_size)
            weight = F.softmax(weight / math.sqrt(key.size(1)), -1)
            return WeightingFunction.apply(value, weight, self.kernel_size, self.kernel_size)
        with torch.no_grad():
            query = self.query_project(target_feats.contiguous())
            key = self.key_project(source_feats.contiguous())
            value = self.value_project(source_feats.contiguous())
            return ops.local_attention(query, key, value, self.kernel_size, 1.0 / math.sqrt(key.size(1)))
