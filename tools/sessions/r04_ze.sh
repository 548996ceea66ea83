#!/bin/bash
# round 4 session ze: the whole -m gpu suite as the driver runs it, on the tree with the option-variant fixtures and the two-slot
# preflight
mkdir -p gpurun_out
timeout 420 python -m pytest tests -x -q -m gpu > gpurun_out/r04_ze_pytest_gpu_suite.txt 2>&1
echo "rc=$?" >> gpurun_out/r04_ze_pytest_gpu_suite.txt
tail -6 gpurun_out/r04_ze_pytest_gpu_suite.txt
