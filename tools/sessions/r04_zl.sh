#!/bin/bash
# round 4 session zl: after the rotate-NMS fix (a task without NMS is not cut to pre_maxsize): every NMS test of the op and head suites
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_head_gpu.py tests/test_ops_gpu.py -q -k "nms" > gpurun_out/r04_zl_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04_zl_tests.log
tail -15 gpurun_out/r04_zl_tests.log
