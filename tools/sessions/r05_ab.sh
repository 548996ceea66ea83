#!/bin/bash
# round 5 session ab: PMC FETCH_SIZE / WRITE_SIZE of every kernel of the eager 32-frame step on the final tree (after the re-read diets)
O=$PWD/gpurun_out/r05_ab; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "swapped_operands" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -n 3 $O/tests.log | cut -c1-200
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_l_$C -o p -- python $R/bench.py --graph off --steps 3 --warmup 1 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/pmc_l_$C.json 2> $O/pmc_l_$C.err )
  python tools/pmc_summary.py $(find $O/pmc_l_$C -name '*_results.db' | head -1) > $O/pmc_l_$C.txt 2>&1
  rm -rf $O/pmc_l_$C
  head -16 $O/pmc_l_$C.txt | cut -c1-150
done
