#!/bin/bash
# round 6 experiment: NCHW-source halo conv, timing variants
O=$PWD/gpurun_out/r06_nc4; mkdir -p $O
export FF3D_LIB=$PWD/focalformer3d_amd/lib/libff3d_hip_exp.so
for rep in 1; do
  B=32 H=180 W=180 GEO=0 timeout 300 python tools/experiments/exp_halo_nchw.py 2>&1 | tail -5 >> $O/ab.txt
done
cat $O/ab.txt
