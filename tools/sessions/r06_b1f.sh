#!/bin/bash
# round 6, third session: kernel tables of the eager 1-frame and 4-frame steps on the FINAL tree (after rows 11-14 of DESIGN section 0)
O=$PWD/gpurun_out/r06_b1f; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() { name=$1; shift; ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o r -- python $R/bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions "$@" > $O/bench_under_rocprof_$name.json 2> $O/rocprof_$name.err ); DB=$(find $O/prof_$name -name '*_results.db' | head -1); python tools/rocprof_last_step.py $DB 80 > $O/bench_${name}_kernel_stats_last_step.txt 2>&1; rm -rf $O/prof_$name; head -8 $O/bench_${name}_kernel_stats_last_step.txt | cut -c1-150; }
prof b1_eager --graph off --batch 1 --steps 5 --warmup 3
prof b4_eager --graph off --batch 4 --steps 5 --warmup 3
