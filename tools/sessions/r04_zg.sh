#!/bin/bash
# round 4 session zg: roi_mlp.0 (19 200 x 37 632 x 512) - split-K sweep on the shipped 128 x 128 kernel and on the 256 x 128
# 8-wave instance of the experiments library
mkdir -p gpurun_out
{
timeout 120 python tools/experiments/exp_roi_mlp.py
FF3D_LIB=$PWD/focalformer3d_amd/lib/libff3d_hip_exp.so FF3D_SPLITMM_VARIANT=4 timeout 120 python tools/experiments/exp_roi_mlp.py
} > gpurun_out/r04_zg_roi_mlp_ab.txt 2>&1
echo "rc=$?"
cat gpurun_out/r04_zg_roi_mlp_ab.txt | tail -8
