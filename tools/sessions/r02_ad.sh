#!/bin/bash
O=$PWD/gpurun_out/r02_ad; mkdir -p $O
export TMPDIR=/tmp
for v in 0 1 0 1; do FF3D_GEMM_WS=$v timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | tail -1 | tee -a $O/ws.txt; done
for v in 0 1; do FF3D_GEMM_WS=$v K=128 N=384 timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | tail -1 | tee -a $O/ws.txt; done
for a in 1 9 10; do echo -n "WS_ABLATE=$a " | tee -a $O/ws.txt; FF3D_WS_ABLATE=$a timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | tail -1 | tee -a $O/ws.txt; done
timeout 900 python -m pytest tests/test_bench_shape_gpu.py -x -q -m gpu -k "gemm" > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gemm.log
