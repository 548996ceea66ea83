#!/bin/bash
# round 6: NCHW-source halo conv, two more request variants (non-temporal; ahead of the weight DMAs)
O=$PWD/gpurun_out/r06_nc9; mkdir -p $O
FF3D_LIB=$PWD/focalformer3d_amd/lib/libff3d_hip_exp.so B=32 H=180 W=180 timeout 300 python tools/experiments/exp_halo_nchw.py 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
