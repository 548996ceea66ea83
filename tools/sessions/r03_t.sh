#!/bin/bash
# round 3 session t: all value projections of the decoder as ONE periodic weight-stationary GEMM over the un-embedded pyramid pair
# (row-bias table = pos_embed @ W^T + b): flatten writes raw + one pair instead of raw + two embedded pairs.  Tests + A/B.
O=$PWD/gpurun_out/r03_t; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "rowbias or gemm" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/pytest_ops.log | cut -c1-400
FF3D_FUSE_VALUE=1 timeout 1200 python -m pytest tests/test_head_gpu.py tests/test_bench_shape_gpu.py tests/test_small_batch_gpu.py -x -q -m gpu > $O/pytest_head_fused.log 2>&1; echo "head (fused value) rc=$?"; tail -3 $O/pytest_head_fused.log | cut -c1-400
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['execution'][:16], {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if k.startswith('gemm')})
PY
}
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-strong-probe "$@" > $O/$name.json 2> $O/$name.err; show $O/$name.json; }
for rep in 1 2; do
FF3D_FUSE_VALUE=1 b bench_b32_fused_$rep
b bench_b32_split_$rep
done
FF3D_FUSE_VALUE=1 FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1_fused --batch 4 --steps 40 --warmup 5
FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1_split --batch 4 --steps 40 --warmup 5
FF3D_FUSE_VALUE=1 b bench_b4_graph_fused --batch 4 --steps 40 --warmup 5
b bench_b4_graph_split --batch 4 --steps 40 --warmup 5
FF3D_FUSE_VALUE=1 b bench_b1_graph_fused --batch 1 --steps 40 --warmup 5
b bench_b1_graph_split --batch 1 --steps 40 --warmup 5
( cd /tmp && FF3D_FUSE_VALUE=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-strong-probe > $O/bench_under_rocprof_b32_fused.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_b32_fused_kernel_stats_last_step.txt 2>&1
find $O/prof_b32 -name '*.db' -delete
head -14 $O/bench_b32_fused_kernel_stats_last_step.txt | cut -c1-150
