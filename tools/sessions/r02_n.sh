#!/bin/bash
O=gpurun_out/r02_n; mkdir -p $O
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q --timeout 1500 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log | cut -c1-300
timeout 600 python bench.py > $O/bench_b32.json 2> $O/bench_b32.err; tail -2 $O/bench_b32.err; cut -c1-200 $O/bench_b32.json
for B in 1 4 8; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_b${B}.json 2> $O/bench_b${B}.err
  cut -c1-200 $O/bench_b${B}.json
done
for B in 32 4; do
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_b$B -o r -- python $GRAFT_REPO_ROOT/bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_b${B}_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof_b$B.err )
DB=$(find $O/prof_b$B -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/b${B}_kernel_stats_last_step.txt 2>&1
python tools/rocprof_summary.py $DB 60 > $O/b${B}_kernel_stats_all.txt 2>&1
done
find $O -name '*.db' -delete
head -30 $O/b32_kernel_stats_last_step.txt | cut -c1-150
