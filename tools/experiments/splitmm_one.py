#!/usr/bin/env python
"""A few launches of the 256-channel split-fp16 conv (B=32, 180x180) - target for rocprofv3 PMC passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from focalformer3d_amd import ops                                  # noqa: E402

torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.randn(B, 256, 180, 180, device='cuda')
w = torch.randn(256, 256, 3, 3, device='cuda') * 0.03
b = torch.randn(256, device='cuda')
xs, ws = ops.split_f16(x, True), ops.split_weight_f16(w)
for _ in range(4):
    y = ops.conv3x3_f16x3(xs, ws, b, True)
torch.cuda.synchronize()
print('ok', float(y.abs().mean()))
