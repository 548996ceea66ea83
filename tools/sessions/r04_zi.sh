#!/bin/bash
# round 4 session zi: the Waymo15-shaped fixture (14 x 14 RoI grid + class-aware regression) produced by the reference, on the HIP path
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_head_gpu.py -q -k "head_opt_classaware" > gpurun_out/r04_zi_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04_zi_tests.log
tail -25 gpurun_out/r04_zi_tests.log
