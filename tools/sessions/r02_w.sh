#!/bin/bash
# round 2 session w: RoI sampler backward kernel (new) - parity, training-step timing before/after, kernel stats
O=$PWD/gpurun_out/r02_w; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_train_forward_gpu.py -x -q -m gpu -k "roi or train" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
timeout 300 python tools/bench_train_step.py 4 256 > $O/train_step_b4_c256.json 2> $O/err1; cat $O/train_step_b4_c256.json
timeout 300 python tools/bench_train_step.py 4 128 > $O/train_step_b4_c128.json 2> $O/err2; cat $O/train_step_b4_c128.json
timeout 300 python tools/bench_train_step.py 2 256 > $O/train_step_b2_c256.json 2> $O/err3; cat $O/train_step_b2_c256.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_train -o r -- python $GRAFT_REPO_ROOT/tools/bench_train_step.py 4 256 > $O/train_under_rocprof.json 2> $O/rocprof_train.err )
DB=$(find $O/prof_train -name '*_results.db' | head -1)
python tools/rocprof_summary.py $DB 45 > $O/train_step_kernel_stats.txt 2>&1; head -36 $O/train_step_kernel_stats.txt | cut -c1-150
find $O -name '*.db' -delete
