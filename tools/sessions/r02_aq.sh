#!/bin/bash
O=$PWD/gpurun_out/r02_aq; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_train_forward_gpu.py -x -q -m gpu -k "msda or train" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 300 python tools/bench_train_step.py 4 256 > $O/train_step_b4_c256.json 2> $O/err1; cat $O/train_step_b4_c256.json
timeout 300 python tools/bench_train_step.py 4 128 > $O/train_step_b4_c128.json 2> $O/err2; cat $O/train_step_b4_c128.json
