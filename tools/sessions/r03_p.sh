#!/bin/bash
# round 3 session p: training step again (session o's box gave a 10 ms slower forward at C = 128 than session j's): bench twice + kernel stats
O=$PWD/gpurun_out/r03_p; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do timeout 600 python tools/bench_train_step.py > $O/train_step_$i.json 2> $O/train_step_$i.err; tail -1 $O/train_step_$i.json | cut -c1-300; done
C=256 timeout 600 python tools/bench_train_step.py 4 256 > $O/train_step_c256.json 2> $O/train_step_c256.err; tail -1 $O/train_step_c256.json | cut -c1-300
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_train -o r -- python $R/tools/bench_train_step.py > $O/train_step_under_rocprof.json 2> $O/rocprof_train.err )
DB=$(find $O/prof_train -name '*_results.db' | head -1)
python tools/rocprof_summary.py $DB 40 > $O/train_step_kernel_stats.txt 2>&1
find $O -name '*.db' -delete
head -30 $O/train_step_kernel_stats.txt | cut -c1-150
