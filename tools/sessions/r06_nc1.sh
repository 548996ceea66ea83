#!/bin/bash
# round 6 experiment: halo conv over the caller's NCHW fp32 map (conversion folded into the halo staging) vs conversion pass + conv
O=$PWD/gpurun_out/r06_nc1; mkdir -p $O
export FF3D_LIB=$PWD/focalformer3d_amd/lib/libff3d_hip_exp.so
for rep in 1 2; do
  B=32 H=180 W=180 GEO=0 timeout 300 python tools/experiments/exp_halo_nchw.py 2>&1 | tail -4 >> $O/ab.txt
done
B=8 H=468 W=468 GEO=1 FF3D_HALO_TAP2=0 timeout 300 python tools/experiments/exp_halo_nchw.py 2>&1 | tail -4 >> $O/ab.txt
B=8 H=468 W=468 GEO=1 timeout 300 python tools/experiments/exp_halo_nchw.py 2>&1 | tail -4 >> $O/ab.txt
cat $O/ab.txt
