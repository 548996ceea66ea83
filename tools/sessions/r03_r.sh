#!/bin/bash
# round 3 session r: value path on a side stream under the heatmap stages (<= 4 frames): tests, A/B against FF3D_OVERLAP_VALUE_MAX_B=0
O=$PWD/gpurun_out/r03_r; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_small_batch_gpu.py tests/test_head_gpu.py tests/test_bench_shape_gpu.py -x -q -m gpu > $O/pytest_head.log 2>&1; echo "head rc=$?"; tail -3 $O/pytest_head.log | cut -c1-300
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['execution'][:16])
PY
}
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-strong-probe "$@" > $O/$name.json 2> $O/$name.err; show $O/$name.json; }
for rep in 1 2; do
b bench_b1_graph_overlap_$rep --batch 1 --steps 40 --warmup 5
FF3D_OVERLAP_VALUE_MAX_B=0 b bench_b1_graph_serial_$rep --batch 1 --steps 40 --warmup 5
done
b bench_b2_graph_overlap --batch 2 --steps 40 --warmup 5
FF3D_OVERLAP_VALUE_MAX_B=0 b bench_b2_graph_serial --batch 2 --steps 40 --warmup 5
b bench_b4_graph_overlap --batch 4 --steps 40 --warmup 5
FF3D_OVERLAP_VALUE_MAX_B=0 b bench_b4_graph_serial --batch 4 --steps 40 --warmup 5
FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1_overlap --batch 4 --steps 40 --warmup 5
FF3D_OVERLAP_VALUE_MAX_B=0 FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1_serial --batch 4 --steps 40 --warmup 5
FF3D_OVERLAP_VALUE_MAX_B=8 b bench_b8_overlap --batch 8 --steps 20 --warmup 5
b bench_b8_serial --batch 8 --steps 20 --warmup 5
