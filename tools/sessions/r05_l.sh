#!/bin/bash
# round 5 session l: PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per pass, --kernel-trace only) of the eager steps of the three workloads
O=$PWD/gpurun_out/r05_l; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  for wl in l lc waymo; do
    ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_${wl}_$C -o p -- python $R/bench.py --workload $wl --graph off --steps 3 --warmup 1 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/pmc_${wl}_$C.json 2> $O/pmc_${wl}_$C.err )
    python tools/pmc_summary.py $(find $O/pmc_${wl}_$C -name '*_results.db' | head -1) msda_fwd conv3x3_halo splitmm split_nchw bev_flatten roi_grid linear locatt cam_sample > $O/pmc_${wl}_$C.txt 2>&1
    rm -rf $O/pmc_${wl}_$C
    grep -i "msda\|linear_rows\|locatt\|ws_kernel" $O/pmc_${wl}_$C.txt | head -8 | cut -c1-190
  done
done
