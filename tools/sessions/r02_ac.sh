#!/bin/bash
O=$PWD/gpurun_out/r02_ac; mkdir -p $O
export TMPDIR=/tmp
for v in 0 1; do FF3D_GEMM_WS=$v timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | tail -1 | tee -a $O/ws.txt; done
for n in 128 256 384; do for v in 0 1; do FF3D_GEMM_WS=$v N=$n timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | tail -1 | tee -a $O/ws.txt; done; done
