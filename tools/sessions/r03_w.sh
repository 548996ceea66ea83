#!/bin/bash
# round 3 session w: tile height of the linear kernel by rounds of resident blocks (auto rule vs FF3D_LIN_BM=64 / 32)
O=$PWD/gpurun_out/r03_w; mkdir -p $O
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        lin = {k.split(' ')[1]: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if k.startswith('linear')}
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], lin)
PY
}
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-strong-probe "$@" > $O/$name.json 2> $O/$name.err; show $O/$name.json; }
for rep in 1 2; do
b bench_b32_auto_$rep
FF3D_LIN_BM=64 b bench_b32_bm64_$rep
done
FF3D_LIN_BM=32 b bench_b32_bm32
b bench_b16_auto --batch 16
FF3D_LIN_BM=64 b bench_b16_bm64 --batch 16
FF3D_LIN_BM=32 b bench_b16_bm32 --batch 16
b bench_b8_auto --batch 8 --graph off
FF3D_LIN_BM=32 b bench_b8_bm32 --batch 8 --graph off
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "linear" > $O/pytest_linear.log 2>&1; echo "linear tests rc=$?"; tail -2 $O/pytest_linear.log | cut -c1-300
