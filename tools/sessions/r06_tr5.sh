#!/bin/bash
# round 6: training step after caching the configuration-only device tensors of the training forward (host profile before: r06_tr4)
O=$PWD/gpurun_out/r06_tr5; mkdir -p $O
timeout 1200 python -m pytest tests/test_train_forward_gpu.py tests/test_training_gpu.py -q -m gpu 2>&1 | tail -3 > $O/tests_train.txt
for c in 256 128; do for i in 1 2 3; do timeout 600 python tools/bench_train_step.py 4 $c 2>&1 | grep '^{' >> $O/train_step.txt; done; done
timeout 900 python tools/profile_host_train.py 256 6 > $O/host_profile.txt 2>&1
cat $O/tests_train.txt; cut -c40-245 $O/train_step.txt; grep "^====" $O/host_profile.txt; sed -n 3,14p $O/host_profile.txt | cut -c1-150
