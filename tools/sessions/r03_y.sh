#!/bin/bash
# round 3 session y: kernel lists of the other two BASELINE workloads (lc: camera maps -> I2P neck -> head; waymo: 468x468, bf16 GEMMs)
O=$PWD/gpurun_out/r03_y; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in lc waymo; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$wl -o r -- python $R/bench.py --workload $wl --steps 4 --warmup 2 --graph off --no-cpu-baseline --no-strong-probe > $O/bench_under_rocprof_$wl.json 2> $O/rocprof_$wl.err )
  DB=$(find $O/prof_$wl -name '*_results.db' | head -1)
  python tools/rocprof_last_step.py $DB 50 > $O/bench_${wl}_kernel_stats_last_step.txt 2>&1
  find $O/prof_$wl -name '*.db' -delete
  head -32 $O/bench_${wl}_kernel_stats_last_step.txt | cut -c1-160
done
