#!/bin/bash
# round 6, third session: the stride-2 pyramid convs at 1 and 4 frames - swapped-operand instance (default since this round) vs the 128 x 128 tiles
O=$PWD/gpurun_out/r06_s2b; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() { name=$1; shift; ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o r -- python $R/bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions "$@" > $O/bench_under_rocprof_$name.json 2> $O/rocprof_$name.err ); DB=$(find $O/prof_$name -name '*_results.db' | head -1); python tools/rocprof_last_step.py $DB 80 > $O/bench_${name}_kernel_stats_last_step.txt 2>&1; rm -rf $O/prof_$name; echo "== $name"; grep -n "splitmm_kernel\|splitk_reduce\|kernel time" $O/bench_${name}_kernel_stats_last_step.txt | cut -c1-170; }
for b in 1 4; do
  prof b${b}_swap --graph off --batch $b --steps 5 --warmup 3
  FF3D_CONV_S2_SWAP=0 prof b${b}_noswap --graph off --batch $b --steps 5 --warmup 3
done
