#!/bin/bash
# round 6: per-kernel times of the weight-gradient path (rocprofv3 kernel trace)
O=$PWD/gpurun_out/r06_wg2; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && ONLY_FIRST=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python $R/tools/experiments/exp_wgrad.py > $O/run.txt 2> $O/rocprof.err )
DB=$(find $O/prof -name '*_results.db' | head -1)
python tools/rocprof_summary.py $DB 14 > $O/kernel_stats.txt 2>&1
rm -rf $O/prof
cut -c1-200 $O/kernel_stats.txt; tail -3 $O/run.txt
