#!/bin/bash
# round 3 session j: faster training attention kernels, remaining test fixes, whole suite, training-step profile, B=4 benches
O=$PWD/gpurun_out/r03_j; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_head_gpu.py -q -m gpu -k "masked_self_attention or augmentation" > $O/pytest_new.log 2>&1; echo "new rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_new.log | cut -c1-300 | head -20
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -4 $O/pytest_all.log | cut -c1-300
timeout 600 python tools/bench_train_step.py > $O/train_step.json 2> $O/train_step.err; echo "train bench rc=$?"; tail -1 $O/train_step.json | cut -c1-400
C=256 timeout 600 python tools/bench_train_step.py 4 256 > $O/train_step_c256.json 2> $O/train_step_c256.err; tail -1 $O/train_step_c256.json | cut -c1-400
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_train -o r -- python $GRAFT_REPO_ROOT/tools/bench_train_step.py > $O/train_step_under_rocprof.json 2> $O/rocprof_train.err )
DB=$(find $O/prof_train -name '*_results.db' | head -1)
python tools/rocprof_summary.py $DB 40 > $O/train_step_kernel_stats.txt 2>&1
find $O -name '*.db' -delete
head -16 $O/train_step_kernel_stats.txt | cut -c1-150
FF3D_BENCH_FORCE_DIST=1 timeout 300 python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline --no-strong-probe > $O/bench_b4_rccl1_eager.json 2> $O/bench_b4_rccl1_eager.err; cut -c70-130 $O/bench_b4_rccl1_eager.json
timeout 300 python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline --no-strong-probe --graph off > $O/bench_b4_eager.json 2> $O/bench_b4_eager.err; cut -c70-130 $O/bench_b4_eager.json
