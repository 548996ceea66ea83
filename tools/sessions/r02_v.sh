#!/bin/bash
# round 2 session v: training-mode forward / backward (new), bev_pool vs the reference's own kernel, training-step timing
O=$PWD/gpurun_out/r02_v; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_forward_gpu.py tests/test_training_gpu.py -x -q -m gpu > $O/pytest_train.log 2>&1; echo "train rc=$?"; tail -25 $O/pytest_train.log
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "bev_pool or msda_backward" > $O/pytest_bevpool.log 2>&1; echo "bevpool rc=$?"; tail -5 $O/pytest_bevpool.log
timeout 300 python tools/bench_train_step.py 4 128 > $O/train_step_b4_c128.json 2> $O/train_step_b4_c128.err; tail -3 $O/train_step_b4_c128.err; cat $O/train_step_b4_c128.json
timeout 300 python tools/bench_train_step.py 4 256 > $O/train_step_b4_c256.json 2> $O/train_step_b4_c256.err; cat $O/train_step_b4_c256.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_train -o r -- python $GRAFT_REPO_ROOT/tools/bench_train_step.py 4 256 > $O/train_under_rocprof.json 2> $O/rocprof_train.err )
DB=$(find $O/prof_train -name '*_results.db' | head -1)
python tools/rocprof_summary.py $DB 40 > $O/train_step_kernel_stats.txt 2>&1; head -30 $O/train_step_kernel_stats.txt
find $O -name '*.db' -delete
