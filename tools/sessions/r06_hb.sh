#!/bin/bash
# round 6: first GPU run of the heatmap_box branch (thin task heads, query boxes, 'boxcls' mask): its tests + the head suites
O=$PWD/gpurun_out/r06_hb; mkdir -p $O
timeout 900 python -m pytest tests/test_heatbox_gpu.py -q -m gpu -x 2>&1 | tail -25 > $O/heatbox.log
cat $O/heatbox.log
timeout 900 python -m pytest tests/test_head_gpu.py tests/test_round6_gpu.py -q -m gpu 2>&1 | tail -8 > $O/head.log
cat $O/head.log
