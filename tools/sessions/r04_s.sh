#!/bin/bash
# round 4 session s: kernel profile of the training step at C = 256 (where do the 52 ms go?)
O=$PWD/gpurun_out/r04_s; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 python tools/bench_train_step.py 4 256 > $O/train_step_c256.json 2> $O/train_step_c256.err; tail -1 $O/train_step_c256.json | cut -c1-400
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_train -o r -- python $R/tools/bench_train_step.py 4 256 > $O/train_under_rocprof.json 2> $O/rocprof_train.err )
DB=$(find $O/prof_train -name '*_results.db' | head -1)
python tools/rocprof_summary.py $DB 45 > $O/train_step_c256_kernel_stats.txt 2>&1; find $O/prof_train -name '*.db' -delete
head -50 $O/train_step_c256_kernel_stats.txt | cut -c1-210
