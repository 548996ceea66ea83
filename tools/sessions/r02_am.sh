#!/bin/bash
O=$PWD/gpurun_out/r02_am; mkdir -p $O
export TMPDIR=/tmp
for nw in 4 8 4 8; do FF3D_ATTN_NW=$nw timeout 100 python tools/experiments/exp_attn.py 2>&1 | tail -1 | tee -a $O/attn.txt; done
for b in 1 4; do for nw in 4 8; do B=$b FF3D_ATTN_NW=$nw timeout 100 python tools/experiments/exp_attn.py 2>&1 | tail -1 | tee -a $O/attn.txt; done; done
NQ=693 timeout 100 python tools/experiments/exp_attn.py 2>&1 | tail -1 | tee -a $O/attn.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or attn" > $O/pytest_attn.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_attn.log
