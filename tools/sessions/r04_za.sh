#!/bin/bash
# round 4 session za: effective clock and MFMA-busy (in cycles) of EVERY kernel of the eager 32-frame step
O=$PWD/gpurun_out/r04_za; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do
  ( cd /tmp && timeout 400 rocprofv3 --pmc $C --kernel-trace --stats -d $O/p_$C -o p -- python $R/bench.py --graph off --steps 3 --warmup 1 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_$C.json 2> $O/bench_$C.err )
  DB=$(find $O/p_$C -name '*_results.db' | head -1)
  python tools/pmc_summary.py $DB > $O/pmc_$C.txt 2>&1
  python tools/rocprof_summary.py $DB 30 > $O/stats_$C.txt 2>&1
  find $O/p_$C -name '*.db' -delete
done
head -14 $O/pmc_GRBM_GUI_ACTIVE.txt | cut -c1-170
