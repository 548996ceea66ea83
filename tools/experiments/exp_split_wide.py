"""Round 6 (third session): the wide form of the transposing split (64 pixels x all channels per block, next tile's requests in flight
under the stores; splitmm.hip) against the 64 x 64 form.  Run once per form (the switch is read at library load):
    python tools/experiments/exp_split_wide.py            ;  FF3D_SPLIT_WIDE=0 python tools/experiments/exp_split_wide.py
Prints per shape the time of one conversion (two launches + the check) and a digest of the planes + exponent: the digests of the two
runs must agree (bit-identical planes)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops  # noqa: E402


def t(fn, n=10, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def digest(p):
    h = hashlib.sha256()
    for a in (p[0], p[1], p.exp):
        h.update(a.contiguous().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


print('FF3D_SPLIT_WIDE =', os.environ.get('FF3D_SPLIT_WIDE', '(default: wide)'))
g = torch.Generator().manual_seed(0)
for (B, Cc, H, W) in ((32, 256, 180, 180), (8, 256, 468, 468), (4, 512, 180, 180), (4, 128, 180, 180), (3, 384, 50, 52), (2, 256, 33, 36),
                      (1, 256, 7, 4), (2, 64, 90, 90), (2, 80, 232, 400)):
    x = (torch.randn(B, Cc, H, W, generator=g) * 3.0).cuda()
    hint = ops.new_hint(x.device)
    p = ops.split_f16(x, to_nhwc=True, hint=hint)             # first call: guess 0 -> redo
    d0 = digest(p)
    p = ops.split_f16(x, to_nhwc=True, hint=hint)             # steady state
    d1 = digest(p)
    ref_hi = (x * 2.0 ** -int(p.exp)).permute(0, 2, 3, 1).half()
    ok = bool(torch.equal(ref_hi, p[0]))
    ms = t(lambda: ops.split_f16(x, to_nhwc=True, hint=hint))
    gb = x.numel() * 8 / 1e9
    line = 'B=%d C=%d %dx%d: %.3f ms (%.2f TB/s of read + write)  digest %s / %s  hi == fp16(x * 2^-e): %s' % (B, Cc, H, W, ms, gb / ms, d0, d1, ok)
    if B * Cc * H * W > 2e8 or Cc == 128:
        xs = [x, (x * 0.5 + 1.0)]
        hs = [ops.new_hint(x.device) for _ in xs]
        ps = ops.split_f16_nhwc_group(xs, hs)
        ps = ops.split_f16_nhwc_group(xs, hs)
        msg = t(lambda: ops.split_f16_nhwc_group(xs, hs))
        line += ' | group of 2: %.3f ms, digests %s %s' % (msg, digest(ps[0]), digest(ps[1]))
    print(line)
