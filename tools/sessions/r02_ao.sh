#!/bin/bash
O=$PWD/gpurun_out/r02_ao; mkdir -p $O
export TMPDIR=/tmp
for v in nopoison default grid_sample per_frame_targets; do timeout 200 python tools/debug_train_poison.py train_step_waymo $v 2>&1 | grep -E "worst|losses" | cut -c1-900 | tee -a $O/poison2.txt; done
