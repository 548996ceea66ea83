#!/bin/bash
O=$PWD/gpurun_out/r02_ae; mkdir -p $O
export TMPDIR=/tmp
for b in 1 2 3 4 8; do for v in 0 1; do FF3D_GEMM_WS_MINM=1 FF3D_GEMM_WS=$v B=$b timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | tail -1 | tee -a $O/ws_small.txt; done; done
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_all.log
