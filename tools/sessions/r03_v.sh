#!/bin/bash
# round 3 session v: split-K slice count of roi_mlp.0 (19200 x 37632 x 512 at 32 frames: 600 tiles on 512 block slots): last-round fill
O=$PWD/gpurun_out/r03_v; mkdir -p $O
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if k.startswith('gemm 19200') or k.startswith('gemm 9600') or k.startswith('gemm 4800')})
PY
}
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-strong-probe "$@" > $O/$name.json 2> $O/$name.err; show $O/$name.json; }
for ks in 2 3 4 5 6 1; do FF3D_GEMM_KSPLIT_FORCE=$ks b bench_b32_ks$ks; done
b bench_b32_rule
b bench_b16_rule --batch 16
FF3D_GEMM_KSPLIT_FORCE=1 b bench_b16_ks1 --batch 16
FF3D_GEMM_KSPLIT_FORCE=2 b bench_b16_ks2 --batch 16
b bench_b8_rule --batch 8 --graph off
FF3D_GEMM_KSPLIT_FORCE=2 b bench_b8_ks2 --batch 8 --graph off
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm or ksplit or dense" > $O/pytest_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -2 $O/pytest_gemm.log | cut -c1-300
