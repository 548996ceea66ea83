#!/bin/bash
# round 6: training step with the own weight-gradient kernel on value_proj (forward_train_bf) - A/B against the framework's, then the
# kernel table of the own run (second process: MIOpen's find results of the first are cached)
O=$PWD/gpurun_out/r06_wg9; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_train_forward_gpu.py tests/test_training_gpu.py -q -m gpu 2>&1 | tail -3 > $O/tests_train.txt
for i in 1 2; do
  timeout 600 python tools/bench_train_step.py 4 256 2>&1 | grep '^{' >> $O/train_step_c256.txt
  FF3D_WGRAD_MIN_ROWS=0 timeout 600 python tools/bench_train_step.py 4 256 2>&1 | grep '^{' >> $O/train_step_c256_vendor_wgrad.txt
done
timeout 600 python tools/bench_train_step.py 4 128 2>&1 | grep '^{' >> $O/train_step_c128.txt
FF3D_WGRAD_MIN_ROWS=0 timeout 600 python tools/bench_train_step.py 4 128 2>&1 | grep '^{' >> $O/train_step_c128_vendor_wgrad.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python $R/tools/bench_train_step.py 4 256 > $O/run_prof.txt 2> $O/rocprof.err )
DB=$(find $O/prof -name '*_results.db' | head -1); python tools/rocprof_summary.py $DB 45 > $O/train_kernel_stats.txt 2>&1; rm -rf $O/prof
cat $O/tests_train.txt; cut -c1-260 $O/train_step_c*.txt; head -16 $O/train_kernel_stats.txt | cut -c1-160
