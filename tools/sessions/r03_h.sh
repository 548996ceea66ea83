#!/bin/bash
# round 3 session h: training attention kernels, fixed augmentation / training-route tests, whole suite (strict), train-step bench
O=$PWD/gpurun_out/r03_h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_head_gpu.py tests/test_training_gpu.py tests/test_train_forward_gpu.py -q -m gpu -k "masked_self_attention or augmentation or training_route or training_step" > $O/pytest_new.log 2>&1; echo "new rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_new.log | cut -c1-300 | head -30
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -4 $O/pytest_all.log | cut -c1-300
timeout 600 python tools/bench_train_step.py > $O/train_step.json 2> $O/train_step.err; echo "train bench rc=$?"; tail -3 $O/train_step.json | cut -c1-600
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_train -o r -- python $GRAFT_REPO_ROOT/tools/bench_train_step.py > $O/train_step_under_rocprof.json 2> $O/rocprof_train.err )
DB=$(find $O/prof_train -name '*_results.db' | head -1)
python tools/rocprof_summary.py $DB 40 > $O/train_step_kernel_stats.txt 2>&1 || python tools/rocprof_last_step.py $DB 40 > $O/train_step_kernel_stats.txt 2>&1
find $O -name '*.db' -delete
head -30 $O/train_step_kernel_stats.txt | cut -c1-150
