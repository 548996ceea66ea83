#!/bin/bash
# round 6, third session: split-K pyramid convs, sweep
O=$PWD/gpurun_out/r06_ks1; mkdir -p $O
timeout 600 python tools/experiments/exp_conv_splitk.py 2>&1 | grep -v amdgpu.ids | tee $O/sweep_swap.txt
FF3D_CONV_S2_SWAP=0 timeout 600 python tools/experiments/exp_conv_splitk.py 2>&1 | grep -v amdgpu.ids | tee $O/sweep_noswap.txt
