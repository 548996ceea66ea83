#!/bin/bash
O=$PWD/gpurun_out/r02_al; mkdir -p $O
export TMPDIR=/tmp
for m in default math; do for n in train_step_waymo train_step_nus; do FF3D_TRAIN_SDPA=$m timeout 200 python tools/debug_train_noise.py $n 2>&1 | grep worst | tee -a $O/noise.txt; done; done
