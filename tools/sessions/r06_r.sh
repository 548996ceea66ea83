#!/bin/bash
# round 6 session r: PMC FETCH_SIZE / WRITE_SIZE of the eager 32-frame step with the swapped-operand stride-2 convs (separate --pmc passes)
O=$PWD/gpurun_out/r06_r; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_l_$C -o p -- python $R/bench.py --graph off --steps 3 --warmup 1 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions > $O/pmc_l_$C.json 2> $O/pmc_l_$C.err )
  python tools/pmc_summary.py $(find $O/pmc_l_$C -name '*_results.db' | head -1) > $O/pmc_l_$C.txt 2>&1
  rm -rf $O/pmc_l_$C
  grep -n "splitmm_kernel\|roi_grid\|msda_fwd\|bev_flatten" $O/pmc_l_$C.txt | cut -c1-170
done
