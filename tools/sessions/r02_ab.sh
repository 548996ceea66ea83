#!/bin/bash
O=$PWD/gpurun_out/r02_ab; mkdir -p $O
export TMPDIR=/tmp
for a in 0 1 2 4 8 6 14 7 9 10 12; do echo -n "WS_ABLATE=$a " | tee -a $O/ws_ablate.txt; FF3D_WS_ABLATE=$a timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | tail -1 | tee -a $O/ws_ablate.txt; done
