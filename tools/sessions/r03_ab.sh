#!/bin/bash
# round 3 session ab: grouped heatmap heads at ONE frame (3 x 270 = 810 blocks of the halo form against three implicit-GEMM launches)
O=$PWD/gpurun_out/r03_ab; mkdir -p $O
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['execution'][:16], {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if k.startswith('conv3x3') and 's1' in k})
PY
}
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-strong-probe "$@" > $O/$name.json 2> $O/$name.err; show $O/$name.json; }
for rep in 1 2; do
FF3D_HEADS_GROUP_MIN_BLOCKS=768 b bench_b1_graph_grouped_$rep --batch 1 --steps 40 --warmup 5
b bench_b1_graph_default_$rep --batch 1 --steps 40 --warmup 5
done
FF3D_HEADS_GROUP_MIN_BLOCKS=768 timeout 600 python -m pytest tests/test_head_gpu.py -x -q -m gpu -k "full_size or golden" > $O/pytest_head.log 2>&1; echo "head rc=$?"; tail -2 $O/pytest_head.log | cut -c1-300
