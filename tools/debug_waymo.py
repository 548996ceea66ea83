"""Check every split-fp16 op of one head forward against an fp32 torch evaluation of the SAME operands (pair.value())."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalformer3d_amd import ops
from focalformer3d_amd.synthetic import build_head_from_cfg, focalformer3d_l_head_cfg, stage_features
C, grid = int(sys.argv[1]), int(sys.argv[2])
hc = focalformer3d_l_head_cfg(C=C, grid=grid, num_proposals=250, stages=4, decoder_stages=2, num_classes=3,
                              dataset='Waymo', ffn=256, hidden_channel_roi=128)
head = build_head_from_cfg(hc, seed=5).cuda()
inputs = stage_features(1, C, grid, 4, seed=6)
dev = [inputs[0].cuda(), [t.cuda() for t in inputs[1]]]

def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))

orig_conv, orig_gemm, orig_small, orig_rb = ops.conv3x3_f16x3, ops.gemm_f16x3, ops.conv3x3_small_f16x3, ops.gemm_f16x3_rowbias
def conv(x, w, bias=None, relu=False, stride=1, split_out=False):
    out = orig_conv(x, w, bias, relu, stride, split_out)
    xv = ops.as_pair(x).value().permute(0, 3, 1, 2)
    wv = ops.as_pair(w).value().permute(0, 3, 1, 2)
    ref = F.conv2d(xv, wv, bias, stride=stride, padding=1)
    ref = ref.relu() if relu else ref
    got = out.value().permute(0, 3, 1, 2) if split_out else out
    print('conv', tuple(xv.shape), '->', tuple(ref.shape), 's', stride, 'pair' if split_out else 'f32', 'rel', rel(got, ref),
          'exps', int(ops.as_pair(x).exp) if ops.as_pair(x).exp is not None else None, int(w.exp),
          (int(out.exp) if split_out else int(out._ff3d_exp)), flush=True)
    return out
def gemm(a, w, bias=None, relu=False, ksplit=None):
    out = orig_gemm(a, w, bias, relu, ksplit)
    ref = ops.as_pair(a).value() @ ops.as_pair(w).value().t() + (bias if bias is not None else 0)
    ref = ref.relu() if relu else ref
    print('gemm', tuple(a[0].shape), tuple(w[0].shape), 'rel', rel(out, ref), 'exps', None if ops.as_pair(a).exp is None else int(ops.as_pair(a).exp), int(w.exp), flush=True)
    return out
def small(x, w, bias, K):
    out = orig_small(x, w, bias, K)
    ref = F.conv2d(ops.as_pair(x).value().permute(0, 3, 1, 2), ops.as_pair(w).value().permute(0, 3, 1, 2)[:K], bias, padding=1)
    print('tail', tuple(ref.shape), 'rel', rel(out, ref), flush=True)
    return out
def rowbias(a, w, table, nb):
    out = orig_rb(a, w, table, nb)
    ref = ops.as_pair(a).value() @ ops.as_pair(w).value().t() + table.repeat(nb, 1)
    print('rowbias', tuple(a[0].shape), tuple(w[0].shape), 'rel', rel(out, ref), flush=True)
    return out
ops.conv3x3_f16x3, ops.gemm_f16x3, ops.conv3x3_small_f16x3, ops.gemm_f16x3_rowbias = conv, gemm, small, rowbias
out = head(dev, None, [{}])[0][0]
ref_head = build_head_from_cfg(hc, seed=5).cuda(); ref_head.set_dense_mode('vendor')
ops.conv3x3_f16x3, ops.gemm_f16x3, ops.conv3x3_small_f16x3, ops.gemm_f16x3_rowbias = orig_conv, orig_gemm, orig_small, orig_rb
ref = ref_head(dev, None, [{}])[0][0]
print({k: float((out[k] - ref[k]).abs().max()) for k in ('center', 'height', 'dim', 'rot', 'heatmap')})
