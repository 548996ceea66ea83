#!/bin/bash
O=$PWD/gpurun_out/r02_ar; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_train -o r -- python $GRAFT_REPO_ROOT/tools/bench_train_step.py 4 256 > $O/train_under_rocprof.json 2> $O/rocprof_train.err )
DB=$(find $O/prof_train -name '*_results.db' | head -1)
python tools/rocprof_summary.py $DB 40 > $O/train_step_kernel_stats.txt 2>&1; head -24 $O/train_step_kernel_stats.txt | cut -c1-140
find $O -name '*.db' -delete
timeout 200 python tools/bench_train_step.py 4 128 2>/dev/null | tail -1
