#!/bin/bash
# round 6: kernel time of the training step with the own weight-gradient kernel and with the framework's (rocprofv3 kernel trace)
O=$PWD/gpurun_out/r06_wg8; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() { name=$1; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o r -- python $R/tools/bench_train_step.py 4 256 > $O/run_$name.txt 2> $O/rocprof_$name.err ); DB=$(find $O/prof_$name -name '*_results.db' | head -1); python tools/rocprof_summary.py $DB 40 > $O/train_${name}_kernel_stats.txt 2>&1; rm -rf $O/prof_$name; head -12 $O/train_${name}_kernel_stats.txt | cut -c1-170; grep '^{' $O/run_$name.txt | cut -c1-250; }
prof own
FF3D_WGRAD_MIN_ROWS=0 prof vendor
