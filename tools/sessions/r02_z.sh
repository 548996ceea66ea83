#!/bin/bash
O=$PWD/gpurun_out/r02_z; mkdir -p $O
export TMPDIR=/tmp
for a in 0 4 12 20 8 16 13 5; do echo -n "ABLATE=$a " | tee -a $O/ablate.txt; FF3D_HALO_PP=0 FF3D_HALO_ABLATE=$a timeout 200 python tools/experiments/exp_halo.py 2>&1 | tail -1 | tee -a $O/ablate.txt; done
