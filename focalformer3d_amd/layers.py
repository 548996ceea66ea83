"""Small building blocks of the head with the reference's parameter layout.

ConvModule / build_conv_layer : mmcv bricks the reference head is written against (FD:14, FD:151-221)
MLP                            : projects/mmdet3d_plugin/models/utils/utils.py:16-28
FFN                            : per-attribute prediction heads, .../models/utils/decoder_utils.py:495-578
gen_sineembed_for_position     : .../models/utils/utils.py:40-53 on the HIP kernel
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

def weight_signature(tensors):
    """Version key of a set of weights: (storage address, in-place version counter) per tensor.  Every weight-derived
    cache in this package (folded BatchNorms, split-fp16 planes, fused projection matrices, cached BEV positional
    embeddings) is keyed on it, so checkpoint loads that bypass torch's load_state_dict hooks (mmcv ``load_checkpoint``
    recursing through ``_load_from_state_dict``), EMA swaps and ``param.data.copy_`` all rebuild the caches."""
    return tuple((t.data_ptr(), t._version) for t in tensors)


_CONV = {'Conv1d': nn.Conv1d, 'Conv2d': nn.Conv2d, None: nn.Conv2d}
_NORM = {'BN1d': nn.BatchNorm1d, 'BN2d': nn.BatchNorm2d, 'BN': nn.BatchNorm2d}


def build_conv_layer(cfg, *args, **kwargs):
    """mmcv ``build_conv_layer``: a bare nn.ConvNd (so ``bias='auto'`` is truthy -> biased conv, FD:213-220)."""
    layer_type = 'Conv2d' if cfg is None else cfg.get('type', 'Conv2d')
    return _CONV[layer_type](*args, **kwargs)


class ConvModule(nn.Module):
    """mmcv ``ConvModule``: conv -> norm -> ReLU with sub-modules named conv / bn / activate;
    ``bias='auto'`` means "bias iff no norm"."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias='auto', conv_cfg=None,
                 norm_cfg=None, act_cfg=dict(type='ReLU'), **kwargs):
        super().__init__()
        self.with_norm = norm_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        ctype = 'Conv2d' if conv_cfg is None else conv_cfg.get('type', 'Conv2d')
        self.conv = _CONV[ctype](in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias)
        if self.with_norm:
            self.bn = _NORM[norm_cfg['type']](out_channels)
        self.with_activation = act_cfg is not None
        if self.with_activation:
            self.activate = nn.ReLU(inplace=True)

    def folded(self):
        """Inference form: BatchNorm (running statistics) folded into the conv -> (weight, bias)."""
        w, b = self.conv.weight, self.conv.bias
        if not self.with_norm:
            return w, b
        bn = self.bn
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        shape = (-1,) + (1,) * (w.dim() - 1)
        w2 = w * scale.view(shape)
        b2 = bn.bias - bn.running_mean * scale if b is None else (b - bn.running_mean) * scale + bn.bias
        return w2.contiguous(), b2.contiguous()

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = self.bn(x)
        return self.activate(x) if self.with_activation else x


class MLP(nn.Module):
    """utils.py:16-28."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        if self.training and x.is_cuda and torch.is_grad_enabled():   # training: over the BEV positions (42 525 rows) the weight gradients take the
            from .autograd import train_linear          # own TN kernel (autograd.train_linear; small inputs: the framework's op)
            for i, layer in enumerate(self.layers):
                x = train_linear(x, layer.weight, layer.bias)
                if i < self.num_layers - 1:
                    x = F.relu(x)
            return x
        for i, layer in enumerate(self.layers):
            x = F.relu(layer(x)) if i < self.num_layers - 1 else layer(x)
        return x


_DIM_T = {}


def sine_dim_t(device):
    """utils.py:44-45, computed once per device by the host framework's pow so the table is identical
    to the reference's."""
    key = str(device)
    if key not in _DIM_T:
        d = torch.arange(128, dtype=torch.float32)
        _DIM_T[key] = (10000 ** (2 * (d // 2) / 128)).to(device)
    return _DIM_T[key]


def gen_sineembed_for_position(pos_tensor, W=1.0, H=1.0):
    """utils.py:40-53 for (..., 2) positions; ``W``/``H`` fuse the FD:869 ``pos / (W, H)`` normalisation."""
    if pos_tensor.size(-1) != 2:
        raise ValueError('Unknown pos_tensor shape(-1):{}'.format(pos_tensor.size(-1)))
    return ops.sine_embed(pos_tensor.contiguous(), sine_dim_t(pos_tensor.device), W, H)


class FFN(nn.Module):
    """decoder_utils.py:495-578: one small Conv1d stack per predicted attribute.  ``forward`` returns
    the same dict; the inference fast path fuses all heads into two GEMMs (``fused_weights``)."""

    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, init_bias=-2.19, conv_cfg=dict(type='Conv1d'),
                 norm_cfg=dict(type='BN1d'), bias='auto', **kwargs):
        super().__init__()
        if final_kernel != 1:
            raise NotImplementedError('final_kernel=1 in every FocalFormer3D config')
        self.heads, self.init_bias, self.head_conv, self.in_channels = heads, init_bias, head_conv, in_channels
        for head in self.heads:
            classes, num_conv = self.heads[head]
            layers, c_in = [], in_channels
            for _ in range(num_conv - 1):
                layers.append(ConvModule(c_in, head_conv, kernel_size=1, stride=1, padding=0, bias=bias,
                                         conv_cfg=conv_cfg, norm_cfg=norm_cfg))
                c_in = head_conv
            layers.append(build_conv_layer(conv_cfg, c_in, classes, kernel_size=1, stride=1, padding=0, bias=True))
            self.__setattr__(head, nn.Sequential(*layers))

    def init_weights(self):
        for head in self.heads:
            if head == 'heatmap':
                self.__getattr__(head)[-1].bias.data.fill_(self.init_bias)

    def fused_weights(self):
        """All heads as two GEMMs: W1 (sum hidden, C) + b1 [BN folded], then a block-diagonal W2 (sum n,
        sum hidden) + b2.  Only for the 2-conv heads every config uses."""
        w1, b1, blocks, b2, sizes = [], [], [], [], []
        for head in self.heads:
            seq = self.__getattr__(head)
            if len(seq) != 2:
                return None
            w, b = seq[0].folded()
            w1.append(w.squeeze(-1))
            b1.append(b)
            blocks.append(seq[1].weight.squeeze(-1))
            b2.append(seq[1].bias)
            sizes.append(seq[1].weight.shape[0])
        W2 = torch.block_diag(*blocks)
        return torch.cat(w1).contiguous(), torch.cat(b1).contiguous(), W2.contiguous(), torch.cat(b2).contiguous(), sizes

    def forward(self, x):
        return {head: self.__getattr__(head)(x) for head in self.heads}
