#!/bin/bash
# round 2, GPU session a: parity at the benchmarked shape, default bench, small-batch profile, 2-rank rehearsal
set -x
O=gpurun_out/r02_a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_bench_shape_gpu.py "tests/test_head_gpu.py::test_head_full_size_vs_oracle" -m gpu -q -x --timeout 1200 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench_b32.json 2> $O/bench_b32.err
for B in 1 4 8; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline --graph on > $O/bench_b${B}_graph.json 2> $O/bench_b${B}_graph.err
  timeout 300 python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline --graph off > $O/bench_b${B}_eager.json 2> $O/bench_b${B}_eager.err
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_b4 -o r -- python $GRAFT_REPO_ROOT/bench.py --batch 4 --steps 5 --warmup 2 --no-cpu-baseline --graph off > $GRAFT_REPO_ROOT/$O/bench_b4_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof_b4.err )
python tools/rocprof_last_step.py $(find $O/prof_b4 -name '*_results.db' | head -1) 60 > $O/b4_kernel_stats_last_step.txt 2>&1
python tools/rocprof_summary.py $(find $O/prof_b4 -name '*_results.db' | head -1) 60 > $O/b4_kernel_stats_all.txt 2>&1
FF3D_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --batch 8 --steps 5 --warmup 2 > $O/bench_gloo2_weak.json 2> $O/bench_gloo2_weak.err
FF3D_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --global-batch 32 --steps 5 --warmup 2 > $O/bench_gloo2_strong.json 2> $O/bench_gloo2_strong.err
rm -rf $O/prof_b4/*/*.db 2>/dev/null; find $O -name '*.db' -size +20M -delete
tail -5 $O/pytest.log; cat $O/bench_b32.json | cut -c1-600; for f in $O/bench_b*_*.json $O/bench_gloo2*.json; do echo $f; cut -c1-200 $f; done
