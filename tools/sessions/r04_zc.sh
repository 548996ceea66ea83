#!/bin/bash
# round 4 session zc: the three option-variant head goldens (class-aware regression, 'pos' mask mode, single-scale value)
# produced by the reference itself, on the HIP path
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_head_gpu.py -q -k "head_opt or option_variants" > gpurun_out/r04_zc_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04_zc_tests.log
tail -5 gpurun_out/r04_zc_tests.log
