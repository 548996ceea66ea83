"""Diff the intermediate tensors of the default (split-fp16) head against the vendor-fp32 head."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalformer3d_amd.synthetic import build_head_from_cfg, focalformer3d_l_head_cfg, stage_features
C, grid = int(sys.argv[1]), int(sys.argv[2])
hc = focalformer3d_l_head_cfg(C=C, grid=grid, num_proposals=250, stages=4, decoder_stages=2, num_classes=3,
                              dataset='Waymo', ffn=256, hidden_channel_roi=128)
a = build_head_from_cfg(hc, seed=5).cuda(); b = build_head_from_cfg(hc, seed=5).cuda(); b.set_dense_mode('vendor')
if len(sys.argv) > 3:
    a.fuse_value_proj = False
inputs = stage_features(1, C, grid, 4, seed=6)
dev = [inputs[0].cuda(), [t.cuda() for t in inputs[1]]]
a._taps, b._taps = {}, {}
oa = a(dev, None, [{}])[0][0]; ob = b(dev, None, [{}])[0][0]
for k in a._taps:
    if k in b._taps and a._taps[k].shape == b._taps[k].shape:
        x, y = a._taps[k].float(), b._taps[k].float()
        print(f'{k:12s} shape {tuple(x.shape)} max|a-b| {float((x - y).abs().max()):.3e}  max|b| {float(y.abs().max()):.3e}  finite {bool(torch.isfinite(x).all())}', flush=True)
    else:
        print(k, 'only in a' if k not in b._taps else 'shape differs', tuple(a._taps[k].shape))
print({k: float((oa[k] - ob[k]).abs().max()) for k in ('center', 'height', 'dim', 'rot', 'heatmap')})
qa, qb = a._taps['qfeat0'], b._taps['qfeat0']
bad = ((qa - qb).abs().amax(-1) > 1e-4)[0]
print('queries with different initial features:', int(bad.sum()), 'of', bad.numel(), 'indices', bad.nonzero().flatten().tolist()[:20])
print('labels equal', bool(torch.equal(a.query_labels, b.query_labels)))
for i, (ha, hb) in enumerate(zip(oa['dense_heatmap'], ob['dense_heatmap'])):
    print('dense_heatmap', i, float((ha - hb).abs().max()), float(hb.abs().max()))
good = ~bad
for k in ('center', 'height', 'dim', 'rot', 'heatmap'):
    n = oa[k].shape[-1] // bad.numel()
    g = good.repeat(n)
    print(k, 'max err over unaffected queries', float((oa[k] - ob[k])[..., g].abs().max()))
