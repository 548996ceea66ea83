#!/bin/bash
# round 4 session zh: the registry's TransFusionBBoxCoder.decode against the reference coder's own outputs (threshold 0.0 and a
# truthy threshold with an empty frame)
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_head_gpu.py -q -k "bbox_coder_decode" > gpurun_out/r04_zh_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04_zh_tests.log
tail -25 gpurun_out/r04_zh_tests.log
