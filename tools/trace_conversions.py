"""Which call sites convert layouts in one eager lc step (BASELINE configs[2]): every ops.split_f16 / split_f16_nhwc_group / unsplit /
nchw_to_nhwc call of one forward with its tensor shape and the two nearest package frames.  Run on the GPU box:
    python tools/trace_conversions.py [B]"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focalformer3d_amd import ops  # noqa: E402
from focalformer3d_amd.runtime import NeckAndHead  # noqa: E402
from focalformer3d_amd.synthetic import build_head_from_cfg, build_neck_from_cfg, focalformer3d_lc_cfgs, lc_inputs  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ncfg, cfg = focalformer3d_lc_cfgs(C=256)
neck = build_neck_from_cfg(ncfg, seed=1, device='cuda')
head = build_head_from_cfg(cfg, seed=0, device='cuda')
img, pts, metas, _ = lc_inputs(B, seed=1, device='cuda')
unit = NeckAndHead(neck, head, metas).eval()
run = lambda: unit.get_bboxes_padded(unit([img, [pts]], None, None))  # noqa: E731
run()
log = collections.Counter()


def wrap(name):
    orig = getattr(ops, name)

    def f(*a, **k):
        x = a[0][0] if isinstance(a[0], (list, tuple)) and not hasattr(a[0], 'exp') else a[0]
        shape = tuple(x[0].shape) if isinstance(x, tuple) else tuple(x.shape)
        fr = [s for s in traceback.extract_stack()[:-1] if 'focalformer3d_amd' in s.filename and not s.filename.endswith('ops.py')][-3:]
        log[(name, shape, str(k.get('to_nhwc', '')), ' <- '.join('%s:%d %s' % (os.path.basename(s.filename), s.lineno, s.name) for s in reversed(fr)))] += 1
        return orig(*a, **k)
    setattr(ops, name, f)


for n in ('split_f16', 'split_f16_nhwc_group', 'unsplit_f16', 'nchw_to_nhwc'):
    if hasattr(ops, n):
        wrap(n)
run()
for k, v in sorted(log.items(), key=lambda kv: -kv[1]):
    print(v, *k)
