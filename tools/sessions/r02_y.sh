#!/bin/bash
# round 2 session y: halo conv timing ablations (which resource bounds the step?)
O=$PWD/gpurun_out/r02_y; mkdir -p $O
export TMPDIR=/tmp
for a in 0 1 2 4 3 5 6 7; do echo -n "ABLATE=$a " | tee -a $O/ablate.txt; FF3D_HALO_PP=0 FF3D_HALO_ABLATE=$a timeout 200 python tools/experiments/exp_halo.py 2>&1 | tail -1 | tee -a $O/ablate.txt; done
