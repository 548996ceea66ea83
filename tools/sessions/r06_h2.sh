#!/bin/bash
# round 6: halo conv with two filter taps per barrier (FF3D_HALO_TAP2=1, 8 x 32 geometry) against the shipped forms
O=$PWD/gpurun_out/r06_h2; mkdir -p $O
run() { timeout 300 python tools/experiments/exp_halo.py 2>&1 | grep "nchw" >> $O/ab.txt; }
for rep in 1 2; do
  B=32 H=180 W=180 run
  B=32 H=180 W=180 FF3D_HALO_GEO=1 run
  B=32 H=180 W=180 FF3D_HALO_GEO=1 FF3D_HALO_TAP2=1 run
  B=48 H=232 W=400 run
  B=48 H=232 W=400 FF3D_HALO_TAP2=1 run
  B=8 H=468 W=468 run
  B=8 H=468 W=468 FF3D_HALO_TAP2=1 run
done
cat $O/ab.txt
