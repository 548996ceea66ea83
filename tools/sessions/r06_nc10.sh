#!/bin/bash
# round 6: NCHW-source halo conv in the 8 x 32 geometry (one tap per barrier), 468 x 468 maps (configs[4]) - experiments library
O=$PWD/gpurun_out/r06_nc10; mkdir -p $O
for i in 1 2; do
FF3D_LIB=$PWD/focalformer3d_amd/lib/libff3d_hip_exp.so PAIR_ONLY=1 B=8 H=468 W=468 timeout 300 python tools/experiments/exp_halo_nchw.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
