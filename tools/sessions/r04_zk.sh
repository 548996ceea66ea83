#!/bin/bash
# round 4 session zk: head.get_bboxes with nms_type None / circle / rotate vs the reference-executed fixtures (both task tables)
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_head_gpu.py -q -k "nms_matches_reference_golden" > gpurun_out/r04_zk_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04_zk_tests.log
tail -40 gpurun_out/r04_zk_tests.log
