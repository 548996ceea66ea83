#!/bin/bash
# round 3 session g: training routes (neck / LSS / local attention), point-cloud augmentation undo, whole suite
O=$PWD/gpurun_out/r03_g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_head_gpu.py tests/test_training_gpu.py -q -m gpu -k "local_context or augmentation or training_route or bev_pool_backward" > $O/pytest_new.log 2>&1; echo "new rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_new.log | cut -c1-300 | head -30
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -4 $O/pytest_all.log | cut -c1-300
timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-strong-probe > $O/bench_b1.json 2> $O/bench_b1.err; cut -c70-130 $O/bench_b1.json
