#!/bin/bash
# round 3 session q: deep-prefetch splitmm instance <2,4> for small grids (1 - 4 frames): tests, A/B against FF3D_SPLITMM_DEEP=0
O=$PWD/gpurun_out/r03_q; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv or gemm or split or dense or halo" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/pytest_ops.log | cut -c1-300
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['execution'][:16], {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if not k.startswith('linear')})
PY
}
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-strong-probe "$@" > $O/$name.json 2> $O/$name.err; show $O/$name.json; }
for rep in 1 2; do
FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1_deep_$rep --batch 4 --steps 40 --warmup 5
FF3D_SPLITMM_DEEP=0 FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1_nodeep_$rep --batch 4 --steps 40 --warmup 5
done
b bench_b4_graph_deep --batch 4 --steps 40 --warmup 5
b bench_b1_graph_deep --batch 1 --steps 40 --warmup 5
FF3D_SPLITMM_DEEP=0 b bench_b1_graph_nodeep --batch 1 --steps 40 --warmup 5
b bench_b2_graph_deep --batch 2 --steps 40 --warmup 5
FF3D_SPLITMM_DEEP=0 b bench_b2_graph_nodeep --batch 2 --steps 40 --warmup 5
b bench_b8_deep --batch 8 --steps 20 --warmup 5
FF3D_SPLITMM_DEEP=0 b bench_b8_nodeep --batch 8 --steps 20 --warmup 5
for B in 4 1; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_b$B -o r -- python $R/bench.py --batch $B --steps 6 --warmup 3 --graph off --no-cpu-baseline --no-strong-probe > $O/bench_under_rocprof_b$B.json 2> $O/rocprof_b$B.err )
  DB=$(find $O/prof_b$B -name '*_results.db' | head -1)
  python tools/rocprof_last_step.py $DB 70 > $O/bench_b${B}_kernel_stats_last_step.txt 2>&1
  find $O/prof_b$B -name '*.db' -delete
  head -12 $O/bench_b${B}_kernel_stats_last_step.txt | cut -c1-150
done
timeout 1200 python -m pytest tests/test_head_gpu.py tests/test_bench_shape_gpu.py -x -q -m gpu > $O/pytest_head.log 2>&1; echo "head rc=$?"; tail -3 $O/pytest_head.log | cut -c1-300
