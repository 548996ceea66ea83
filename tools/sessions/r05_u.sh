#!/bin/bash
# round 5 session u: where does the fused feed-forward kernel wait?  SQ counters of the micro-benchmark, one counter per pass
O=$PWD/gpurun_out/r05_u; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU; do
  ( cd /tmp && timeout 200 rocprofv3 --pmc $C --kernel-trace --stats -d $O/p_$C -o p -- python $R/tools/bench_ffn_rows.py 19200 1024 fused > $O/run_$C.txt 2> $O/run_$C.err )
  DB=$(find $O/p_$C -name '*_results.db' | head -1)
  python tools/pmc_summary.py $DB 2>&1 | grep -i "ffn_rows\|dispatches" > $O/pmc_$C.txt
  find $O/p_$C -name '*.db' -delete
  echo "$C: $(grep ffn_rows $O/pmc_$C.txt | cut -c1-120)"
done
