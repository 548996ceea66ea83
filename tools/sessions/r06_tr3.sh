#!/bin/bash
# round 6: the training step on the round's tree (own weight-gradient kernel + matrix-core attention), three processes per width
O=$PWD/gpurun_out/r06_tr3; mkdir -p $O
for c in 256 128; do for i in 1 2 3; do timeout 600 python tools/bench_train_step.py 4 $c 2>&1 | grep '^{' >> $O/train_step.txt; done; done
FF3D_WGRAD_MIN_ROWS=0 FF3D_MHA_TRAIN_SCALAR=1 timeout 600 python tools/bench_train_step.py 4 256 2>&1 | grep '^{' >> $O/train_step_round5_form.txt
FF3D_WGRAD_MIN_ROWS=0 FF3D_MHA_TRAIN_SCALAR=1 timeout 600 python tools/bench_train_step.py 4 128 2>&1 | grep '^{' >> $O/train_step_round5_form.txt
cut -c40-260 $O/train_step.txt $O/train_step_round5_form.txt
