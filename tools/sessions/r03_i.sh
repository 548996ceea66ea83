#!/bin/bash
# round 3 session i: where does the halo conv's time go?  SQ counters of conv3x3_halo_f16x3_kernel, one counter per pass
# (B=32, 256 -> 256, 180 x 180: tools/experiments/exp_halo.py), for the shipped 4x64 kernel and the 8x64 opt-in kernel
O=$PWD/gpurun_out/r03_i; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for variant in halo4 halo8; do
  [ $variant = halo8 ] && export FF3D_CONV_HALO8=1 || unset FF3D_CONV_HALO8
  timeout 120 python tools/experiments/exp_halo.py 2>&1 | tail -1 | tee -a $O/timing.txt
  for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE; do
    ( cd /tmp && timeout 150 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${variant}_$c -o p -- python $R/tools/experiments/exp_halo.py > $O/pmc_${variant}_$c.log 2>&1 )
    DB=$(find $O/pmc_${variant}_$c -name '*_results.db' | head -1)
    [ -n "$DB" ] && python tools/pmc_summary.py $DB halo | grep -v "^#\|dispatches" | head -3 | cut -c1-170 | tee -a $O/pmc_${variant}.txt
    find $O/pmc_${variant}_$c -name '*.db' -delete
  done
done
