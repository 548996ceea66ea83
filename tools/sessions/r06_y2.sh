#!/bin/bash
# round 6, second session: rocprofv3 kernel tables of the final tree (default command = graph replay; eager 32-frame step) + PMC FETCH / WRITE passes of the eager step
O=$PWD/gpurun_out/r06_y2; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() { name=$1; shift; ( cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o r -- python $R/bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions "$@" > $O/bench_under_rocprof_$name.json 2> $O/rocprof_$name.err ); DB=$(find $O/prof_$name -name '*_results.db' | head -1); python tools/rocprof_last_step.py $DB 60 > $O/bench_${name}_kernel_stats_last_step.txt 2>&1; python tools/rocprof_summary.py $DB 30 > $O/bench_${name}_kernel_stats.txt 2>&1; rm -rf $O/prof_$name; head -6 $O/bench_${name}_kernel_stats_last_step.txt | cut -c1-150; }
prof default
prof b32_eager --graph off --steps 5 --warmup 3
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_l_$C -o p -- python $R/bench.py --graph off --steps 3 --warmup 1 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions > $O/pmc_l_$C.json 2> $O/pmc_l_$C.err )
  python tools/pmc_summary.py $(find $O/pmc_l_$C -name '*_results.db' | head -1) > $O/pmc_l_$C.txt 2>&1
  rm -rf $O/pmc_l_$C
  grep -n "conv3x3_halo\|msda_fwd\|split_nchw" $O/pmc_l_$C.txt | cut -c1-170
done
