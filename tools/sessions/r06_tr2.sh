#!/bin/bash
# round 6: where the GPU idles inside a training step (rocprofv3 kernel trace + tools/rocprof_gaps.py)
O=$PWD/gpurun_out/r06_tr2; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python tools/bench_train_step.py 4 256 > /dev/null 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python $R/tools/bench_train_step.py 4 256 > $O/run_prof.txt 2> $O/rocprof.err )
DB=$(find $O/prof -name '*_results.db' | head -1)
python tools/rocprof_gaps.py $DB multi_tensor_apply 40 > $O/gaps.txt 2>&1
python tools/rocprof_summary.py $DB 70 > $O/train_kernel_stats.txt 2>&1
rm -rf $O/prof
grep '^{' $O/run_prof.txt | cut -c90-250; cat $O/gaps.txt | cut -c1-200
