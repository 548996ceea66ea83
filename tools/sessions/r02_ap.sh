#!/bin/bash
O=$PWD/gpurun_out/r02_ap; mkdir -p $O
export TMPDIR=/tmp
for f in test_bench_shape_gpu test_head_gpu test_ops_gpu; do
  timeout 600 python -m pytest tests/$f.py tests/test_train_forward_gpu.py -q -m gpu -k "not full_size" > $O/$f.log 2>&1; echo "$f + train: rc=$? $(tail -1 $O/$f.log)"; grep -o "grad/[a-z0-9_.]*', [0-9.e-]*" $O/$f.log | head -3
done
