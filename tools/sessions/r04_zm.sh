#!/bin/bash
# round 4 session zm: the single-stage head without the second heatmap (DeformFormer3D_Waymo_L-shaped fixture from the reference)
mkdir -p gpurun_out
timeout 60 python -m pytest tests/test_head_gpu.py -q -k "singleheat" > gpurun_out/r04_zm_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04_zm_tests.log
tail -30 gpurun_out/r04_zm_tests.log
