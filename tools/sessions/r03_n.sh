#!/bin/bash
# round 3 session n: where does the hand-scheduled halo conv lose its time?  de-phased DMA issue (2), ablations: no loop DMA (3),
# no loop barrier (4), neither (5); then the final numbers of the shipped build
O=$PWD/gpurun_out/r03_n; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1 2 3 4 5 2 0; do
  echo -n "FF3D_HALO_PIPE=$v: " | tee -a $O/halo_pipe_variants.txt; FF3D_HALO_PIPE=$v timeout 120 python tools/experiments/exp_halo.py 2>&1 | tail -1 | tee -a $O/halo_pipe_variants.txt
done
FF3D_HALO_PIPE=2 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv or halo" > $O/pytest_conv_pipe2.log 2>&1; echo "conv tests (pipe 2) rc=$?"; tail -2 $O/pytest_conv_pipe2.log | cut -c1-300
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-200 $O/bench_default.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-strong-probe > $O/bench_under_rocprof_b32.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_b32_kernel_stats_last_step.txt 2>&1
find $O/prof_b32 -name '*.db' -delete
head -24 $O/bench_b32_kernel_stats_last_step.txt | cut -c1-150
