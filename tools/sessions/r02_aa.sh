#!/bin/bash
# round 2 session aa: weight-stationary value GEMM - A/B on the same box, parity tests, bench
O=$PWD/gpurun_out/r02_aa; mkdir -p $O
export TMPDIR=/tmp
for v in 0 1 0 1; do FF3D_GEMM_WS=$v timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | tail -1 | tee -a $O/valuegemm_ab.txt; done
for v in 0 1; do FF3D_GEMM_WS=$v K=128 N=384 timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | tail -1 | tee -a $O/valuegemm_ab.txt; done
timeout 900 python -m pytest tests/test_bench_shape_gpu.py -x -q -m gpu -k "gemm" > $O/pytest_gemm.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gemm.log
FF3D_GEMM_WS=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ws0.json 2> $O/bench_ws0.err; cut -c1-120 $O/bench_ws0.json
FF3D_GEMM_WS=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ws1.json 2> $O/bench_ws1.err; cut -c1-120 $O/bench_ws1.json
