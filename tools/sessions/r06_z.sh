#!/bin/bash
# round 6 session z: PMC FETCH_SIZE / WRITE_SIZE of the gather_first step (what the C-wide gather really pulls from HBM)
O=$PWD/gpurun_out/r06_z; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_gf_$C -o p -- python $R/bench.py --value-mode gather_first --graph off --steps 3 --warmup 1 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions > $O/pmc_gf_$C.json 2> $O/pmc_gf_$C.err )
  python tools/pmc_summary.py $(find $O/pmc_gf_$C -name '*_results.db' | head -1) > $O/pmc_gf_$C.txt 2>&1
  rm -rf $O/pmc_gf_$C
  grep -n "msda_fwd\|bev_flatten\|linear_f16x3" $O/pmc_gf_$C.txt | cut -c1-170
done
