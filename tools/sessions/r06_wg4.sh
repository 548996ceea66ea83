#!/bin/bash
# round 6: weight-gradient kernel, fourth form (the unmasked conversion really separate; plain order below 8 slices)
O=$PWD/gpurun_out/r06_wg4; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python tools/experiments/exp_wgrad.py > $O/exp_wgrad.txt 2>&1
( cd /tmp && ONLY_FIRST=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python $R/tools/experiments/exp_wgrad.py > $O/run.txt 2> $O/rocprof.err )
DB=$(find $O/prof -name '*_results.db' | head -1)
python tools/rocprof_summary.py $DB 8 > $O/kernel_stats.txt 2>&1
rm -rf $O/prof
cat $O/exp_wgrad.txt; cut -c1-160 $O/kernel_stats.txt
