#!/bin/bash
# round 2, GPU session h: full GPU suite (no -x) + rocprof of the default bench step
O=gpurun_out/r02_h; mkdir -p $O
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q --timeout 1500 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log | cut -c1-200
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_b32 -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_b32_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/b32_kernel_stats_last_step.txt 2>&1
python tools/rocprof_summary.py $DB 60 > $O/b32_kernel_stats_all.txt 2>&1
find $O -name '*.db' -delete
head -45 $O/b32_kernel_stats_last_step.txt | cut -c1-170
