#!/bin/bash
# round 4 session w: kernel tables of the lc and waymo workloads on the final tree
O=$PWD/gpurun_out/r04_w; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in lc waymo; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$wl -o r -- python $R/bench.py --workload $wl --graph off --steps 5 --warmup 2 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_$wl.json 2> $O/rocprof_$wl.err )
  DB=$(find $O/prof_$wl -name '*_results.db' | head -1)
  python tools/rocprof_last_step.py $DB 40 > $O/bench_${wl}_kernel_stats_last_step.txt 2>&1
  find $O/prof_$wl -name '*.db' -delete
  head -14 $O/bench_${wl}_kernel_stats_last_step.txt | cut -c1-170
done
