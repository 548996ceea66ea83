#!/bin/bash
# round 4 session zj: the FocalFormer3D_LC-shaped neck (LSS camera branch inside) vs the reference-produced fixture, on the HIP path
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_head_gpu.py -q -k "cam_lss" > gpurun_out/r04_zj_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04_zj_tests.log
tail -25 gpurun_out/r04_zj_tests.log
