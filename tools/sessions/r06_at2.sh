#!/bin/bash
# round 6: training attention kernels, per-tensor errors and times (MFMA, then scalar)
O=$PWD/gpurun_out/r06_at2; mkdir -p $O
timeout 300 python tools/experiments/exp_mha_train.py > $O/mfma.txt 2>&1
FF3D_MHA_TRAIN_SCALAR=1 timeout 300 python tools/experiments/exp_mha_train.py > $O/scalar.txt 2>&1
cat $O/mfma.txt; echo; cat $O/scalar.txt
