#!/bin/bash
# round 6: the training route's self-attention on the fp32 matrix cores (csrc/attn_train.hip, *_mfma_kernel) - its test, then the
# training step against the scalar kernels (FF3D_MHA_TRAIN_SCALAR=1), then the kernel table
O=$PWD/gpurun_out/r06_at5; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "masked_self_attention_training" 2>&1 | tail -8 > $O/tests_attn.txt
FF3D_MHA_TRAIN_SCALAR=1 timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "masked_self_attention_training" 2>&1 | tail -3 >> $O/tests_attn.txt
timeout 1200 python -m pytest tests/test_train_forward_gpu.py tests/test_training_gpu.py -q -m gpu 2>&1 | tail -3 > $O/tests_train.txt
for i in 1 2; do
  timeout 600 python tools/bench_train_step.py 4 256 2>&1 | grep '^{' >> $O/train_step_c256_mfma.txt
  FF3D_MHA_TRAIN_SCALAR=1 timeout 600 python tools/bench_train_step.py 4 256 2>&1 | grep '^{' >> $O/train_step_c256_scalar.txt
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python $R/tools/bench_train_step.py 4 256 > $O/run_prof.txt 2> $O/rocprof.err )
DB=$(find $O/prof -name '*_results.db' | head -1); python tools/rocprof_summary.py $DB 60 > $O/train_kernel_stats.txt 2>&1; rm -rf $O/prof
cat $O/tests_attn.txt $O/tests_train.txt; cut -c90-250 $O/train_step_c*.txt; grep "mha_\|^# kernels" $O/train_kernel_stats.txt | cut -c1-150
