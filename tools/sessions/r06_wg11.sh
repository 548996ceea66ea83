#!/bin/bash
# round 6: kernel table of the training step with the batched value projection (why is its forward slow?)
O=$PWD/gpurun_out/r06_wg11; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python tools/bench_train_step.py 4 256 > /dev/null 2>&1     # (MIOpen find results of a first process)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python $R/tools/bench_train_step.py 4 256 > $O/run_prof.txt 2> $O/rocprof.err )
DB=$(find $O/prof -name '*_results.db' | head -1); python tools/rocprof_summary.py $DB 30 > $O/train_kernel_stats.txt 2>&1; rm -rf $O/prof
grep '^{' $O/run_prof.txt | cut -c90-250; head -20 $O/train_kernel_stats.txt | cut -c1-180
