#!/bin/bash
# round 3 session ac: grouped NCHW -> NHWC-pair conversion of the head's input maps: tests, A/B at 1 / 4 / 32 frames
O=$PWD/gpurun_out/r03_ac; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_small_batch_gpu.py tests/test_head_gpu.py tests/test_bench_shape_gpu.py -x -q -m gpu > $O/pytest_head.log 2>&1; echo "head rc=$?"; tail -3 $O/pytest_head.log | cut -c1-400
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "split" > $O/pytest_split.log 2>&1; echo "split rc=$?"; tail -2 $O/pytest_split.log | cut -c1-300
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
FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1_grouped_$rep --batch 4 --steps 40 --warmup 5
FF3D_INPUT_SPLIT_GROUPED=0 FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1_single_$rep --batch 4 --steps 40 --warmup 5
done
b bench_b1_graph_grouped --batch 1 --steps 40 --warmup 5
FF3D_INPUT_SPLIT_GROUPED=0 b bench_b1_graph_single --batch 1 --steps 40 --warmup 5
b bench_b4_graph_grouped --batch 4 --steps 40 --warmup 5
b bench_b32_grouped
FF3D_INPUT_SPLIT_GROUPED=0 b bench_b32_single
