#!/bin/bash
# round 6, third session: wide transposing split vs the 64 x 64 form (digests must agree)
O=$PWD/gpurun_out/r06_sw1; mkdir -p $O
timeout 300 python tools/experiments/exp_split_wide.py 2>&1 | grep -v amdgpu.ids | tee $O/wide.txt
FF3D_SPLIT_WIDE=0 timeout 300 python tools/experiments/exp_split_wide.py 2>&1 | grep -v amdgpu.ids | tee $O/old.txt
