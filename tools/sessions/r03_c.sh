#!/bin/bash
# round 3 session c: new linear kernel + reworked BASELINE config tests + whole suite (strict training-step check) + benches
O=$PWD/gpurun_out/r03_c; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "linear_f16x3" > $O/pytest_linear.log 2>&1; echo "linear rc=$?"; tail -4 $O/pytest_linear.log | cut -c1-300
FF3D_PARITY_STATS=$O/stats timeout 1200 python -m pytest tests/test_baseline_configs_gpu.py -q -m gpu --durations=8 > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_new.log | cut -c1-400 | head -30
for f in $O/stats/*.json; do echo "== $f"; cat $f | tr -d '\n' | cut -c1-1800; echo; done
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_baseline_configs_gpu.py > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -5 $O/pytest_all.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline > $O/bench_b32.json 2> $O/bench_b32.err; echo "b32 rc=$?"; cut -c1-160 $O/bench_b32.json
timeout 300 python bench.py --batch 4 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_b4.json 2> $O/bench_b4.err; cut -c1-160 $O/bench_b4.json
timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_b1.json 2> $O/bench_b1.err; cut -c1-160 $O/bench_b1.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_b32_under_rocprof.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/b32_kernel_stats_last_step.txt 2>&1
find $O -name '*.db' -delete
head -30 $O/b32_kernel_stats_last_step.txt | cut -c1-150
