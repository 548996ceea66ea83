#!/bin/bash
# round 2 session x: halo conv ping-pong schedule A/B (same box), parity tests of the conv paths, default bench
O=$PWD/gpurun_out/r02_x; mkdir -p $O
export TMPDIR=/tmp
for v in 0 1 0 1; do FF3D_HALO_PP=$v timeout 200 python tools/experiments/exp_halo.py 2>&1 | tail -1 | tee -a $O/halo_ab.txt; done
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv or halo or split" > $O/pytest_conv.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_conv.log
FF3D_HALO_PP=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_pp0.json 2> $O/bench_pp0.err; cut -c1-120 $O/bench_pp0.json
FF3D_HALO_PP=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_pp1.json 2> $O/bench_pp1.err; cut -c1-120 $O/bench_pp1.json
