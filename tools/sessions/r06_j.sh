#!/bin/bash
# round 6 session j: local attention on the matrix cores - first correctness run + timing against the scalar kernel
O=$PWD/gpurun_out/r06_j; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_round6_gpu.py -x -q -k "local_attention" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -n "assert\|Error\|passed\|failed" $O/tests.log | head -20
timeout 300 python - > $O/timing.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from focalformer3d_amd import ops
def t(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B, C, H, W = 8, 256, 180, 180
g = torch.Generator(device='cuda').manual_seed(0)
q, k, v = (torch.randn(B, C, H, W, device='cuda', generator=g) for _ in range(3))
rows = lambda x: ops.split_f16(x, to_nhwc=True).map(lambda p_: p_.reshape(B * H * W, C))
qp, kp, vp = rows(q), rows(k), rows(v)
print('scalar fp32 kernel (NCHW): %.3f ms' % t(lambda: ops.local_attention(q, k, v, 9, C ** -0.5)))
print('MFMA pair kernel (pre-pass + attention): %.3f ms' % t(lambda: ops.local_attention_pair(qp, kp, vp, B, H, W, 9, C ** -0.5)))
PY
cat $O/timing.txt | grep -v amdgpu
