#!/bin/bash
# round 6 session m: value mode 'gather_first' - unit + full-size tests, first timing
O=$PWD/gpurun_out/r06_m; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_round6_gpu.py tests/test_ops_gpu.py -x -q -k "gather_rows or gather_first or msda" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -n "Error\|assert \|passed\|failed" $O/tests.log | head -20
