#!/bin/bash
# round 4 session z: effective clock of the halo conv (GRBM_GUI_ACTIVE / kernel time) - the full kernel vs its MFMA-only ablation,
# and MFMA-busy cycles: does the chip clock lower under the full kernel's load (DVFS), and how much of 3.0 ms is that?
O=$PWD/gpurun_out/r04_z; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export FF3D_LIB=$R/focalformer3d_amd/lib/libff3d_hip_exp.so
for abl in 0 6; do
  for C in GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do
    ( cd /tmp && FF3D_HALO_ABLATE=$abl timeout 200 rocprofv3 --pmc $C --kernel-trace --stats -d $O/p_${abl}_$C -o p -- python $R/tools/experiments/exp_halo.py > $O/halo_${abl}_$C.log 2> $O/halo_${abl}_$C.err )
    DB=$(find $O/p_${abl}_$C -name '*_results.db' | head -1)
    python tools/pmc_summary.py $DB conv3x3_halo > $O/pmc_${abl}_$C.txt 2>&1
    python tools/rocprof_summary.py $DB 6 > $O/stats_${abl}_$C.txt 2>&1
    find $O/p_${abl}_$C -name '*.db' -delete
    echo "ABLATE=$abl $C:"; grep -i "halo" $O/pmc_${abl}_$C.txt | head -3 | cut -c1-170; grep -i "halo" $O/stats_${abl}_$C.txt | head -2 | cut -c1-150
  done
done
