#!/bin/bash
# round 4 session zb: neck tests after the last condition change
O=$PWD/gpurun_out/r04_zb; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_head_gpu.py tests/test_baseline_configs_gpu.py tests/test_training_gpu.py -x -q -m gpu -k "neck or lc_chain or encoder or i2p" > $O/pytest_neck.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest_neck.log | cut -c1-300
