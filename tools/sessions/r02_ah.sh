#!/bin/bash
O=$PWD/gpurun_out/r02_ah; mkdir -p $O
export TMPDIR=/tmp
for v in 1 2 1 2; do FF3D_GEMM_WS=$v K=128 N=384 timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | tail -1 | tee -a $O/ws2.txt; done
for v in 1 2; do FF3D_GEMM_WS=$v timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | tail -1 | tee -a $O/ws2.txt; done
for v in 1 2; do FF3D_GEMM_WS=$v K=128 N=768 timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | tail -1 | tee -a $O/ws2.txt; done
