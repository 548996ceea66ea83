#!/bin/bash
# round 4 session zd: the two-slot collective preflight child and the full N > 1 control flow (captured collective, then the
# strong probe as eager steps) in one process
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_bench_cli_gpu.py -q -x > gpurun_out/r04_zd_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04_zd_tests.log
tail -30 gpurun_out/r04_zd_tests.log
