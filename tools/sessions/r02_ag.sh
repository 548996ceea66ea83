#!/bin/bash
O=$PWD/gpurun_out/r02_ag; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_forward_gpu.py tests/test_training_gpu.py -x -q -m gpu > $O/pytest_train.log 2>&1; echo "train rc=$?"; tail -3 $O/pytest_train.log
timeout 300 python tools/bench_train_step.py 4 256 > $O/train_step_b4_c256.json 2> $O/err1; cat $O/train_step_b4_c256.json
timeout 300 python tools/bench_train_step.py 4 128 > $O/train_step_b4_c128.json 2> $O/err2; cat $O/train_step_b4_c128.json
timeout 300 python tools/bench_train_step.py 8 256 > $O/train_step_b8_c256.json 2> $O/err3; cat $O/train_step_b8_c256.json
