#!/bin/bash
# round 6: where the time of linear_wgrad_f16x3_kernel goes (timing ablations of the experiments library; M = 170 100, K = N = 256)
O=$PWD/gpurun_out/r06_wg6; mkdir -p $O
export FF3D_LIB=$PWD/focalformer3d_amd/lib/libff3d_hip_exp.so TIME_ONLY=1
for a in 0 1 2 4 8 3 6 7 14 15 0; do FF3D_WG_ABLATE=$a timeout 120 python tools/experiments/exp_wgrad.py 2>&1 | grep ABLATE; done > $O/ablate.txt
cat $O/ablate.txt
