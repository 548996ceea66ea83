#!/bin/bash
# round 6: first run of the weight-gradient kernel (csrc/wgrad.hip)
mkdir -p gpurun_out/r06_wg1
timeout 600 python tools/experiments/exp_wgrad.py > gpurun_out/r06_wg1/exp_wgrad.txt 2>&1
echo rc=$? >> gpurun_out/r06_wg1/exp_wgrad.txt
tail -20 gpurun_out/r06_wg1/exp_wgrad.txt
