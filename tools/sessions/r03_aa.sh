#!/bin/bash
# round 3 session aa: the S heatmap heads of the multi-stage head in two grouped launches (small batches): tests, A/B
O=$PWD/gpurun_out/r03_aa; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_small_batch_gpu.py tests/test_head_gpu.py tests/test_bench_shape_gpu.py -x -q -m gpu > $O/pytest_head.log 2>&1; echo "head rc=$?"; tail -3 $O/pytest_head.log | cut -c1-400
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
FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1_grouped_$rep --batch 4 --steps 40 --warmup 5
FF3D_HEATMAP_GROUPED=0 FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1_single_$rep --batch 4 --steps 40 --warmup 5
done
b bench_b4_graph_grouped --batch 4 --steps 40 --warmup 5
FF3D_HEATMAP_GROUPED=0 b bench_b4_graph_single --batch 4 --steps 40 --warmup 5
b bench_b2_graph_grouped --batch 2 --steps 40 --warmup 5
FF3D_HEATMAP_GROUPED=0 b bench_b2_graph_single --batch 2 --steps 40 --warmup 5
b bench_b8_grouped --batch 8
FF3D_HEATMAP_GROUPED=0 b bench_b8_single --batch 8
b bench_b1_graph_grouped --batch 1 --steps 40 --warmup 5
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_b4 -o r -- python $GRAFT_REPO_ROOT/bench.py --batch 4 --steps 6 --warmup 3 --graph off --no-cpu-baseline --no-strong-probe > $O/bench_under_rocprof_b4.json 2> $O/rocprof_b4.err )
DB=$(find $O/prof_b4 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 70 > $O/bench_b4_kernel_stats_last_step.txt 2>&1
find $O/prof_b4 -name '*.db' -delete
head -12 $O/bench_b4_kernel_stats_last_step.txt | cut -c1-150
