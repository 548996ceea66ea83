#!/bin/bash
# round 6: training step - the layers' value projections as one linear under autograd (FF3D_TRAIN_BATCH_VALUE_PROJ) on / off
O=$PWD/gpurun_out/r06_wg10; mkdir -p $O
timeout 1200 python -m pytest tests/test_train_forward_gpu.py tests/test_training_gpu.py -q -m gpu 2>&1 | tail -3 > $O/tests_train.txt
for i in 1 2 3; do
  timeout 600 python tools/bench_train_step.py 4 256 2>&1 | grep '^{' >> $O/train_step_c256_batched.txt
  FF3D_TRAIN_BATCH_VALUE_PROJ=0 timeout 600 python tools/bench_train_step.py 4 256 2>&1 | grep '^{' >> $O/train_step_c256_per_layer.txt
done
FF3D_WGRAD_MIN_ROWS=0 timeout 600 python tools/bench_train_step.py 4 256 2>&1 | grep '^{' >> $O/train_step_c256_batched_vendor_wgrad.txt
cat $O/tests_train.txt; cut -c90-250 $O/train_step_c*.txt
