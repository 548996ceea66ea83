#!/bin/bash
# round 6: the weight-gradient kernel wired into the training path (value_proj, the BEV position MLP): its tests, the training-path
# tests, the training step with and without it
O=$PWD/gpurun_out/r06_wg7; mkdir -p $O
timeout 900 python -m pytest tests/test_round6_gpu.py -q -m gpu -k "wgrad or train_linear" 2>&1 | tail -5 > $O/tests_wgrad.txt
timeout 1200 python -m pytest tests/test_train_forward_gpu.py tests/test_training_gpu.py -q -m gpu 2>&1 | tail -5 > $O/tests_train.txt
for c in 256 128; do
  timeout 600 python tools/bench_train_step.py 4 $c 2>&1 | grep '^{' > $O/train_step_c$c.txt
  FF3D_WGRAD_MIN_ROWS=0 timeout 600 python tools/bench_train_step.py 4 $c 2>&1 | grep '^{' > $O/train_step_c${c}_vendor_wgrad.txt
  timeout 600 python tools/bench_train_step.py 4 $c 2>&1 | grep '^{' >> $O/train_step_c$c.txt
  FF3D_WGRAD_MIN_ROWS=0 timeout 600 python tools/bench_train_step.py 4 $c 2>&1 | grep '^{' >> $O/train_step_c${c}_vendor_wgrad.txt
done
cat $O/tests_wgrad.txt $O/tests_train.txt; cut -c1-330 $O/train_step_c*.txt
