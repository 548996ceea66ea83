#!/bin/bash
# round 3 session l: alignment-free halo swizzle (A/B + bank-conflict counter), fused [projection + add + LayerNorm] and q|k|v
# launches (tests, A/B at 4 frames and 1 frame), kernel traces of the 4-frame and 1-frame steps
O=$PWD/gpurun_out/r03_l; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "linear or conv3x3 or halo" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/pytest_ops.log | cut -c1-300
timeout 900 python -m pytest tests/test_head_gpu.py tests/test_bench_shape_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu > $O/pytest_head.log 2>&1; echo "head rc=$?"; tail -3 $O/pytest_head.log | cut -c1-300
for v in new old new old; do
  [ $v = old ] && export FF3D_HALO_ABLATE=16 || unset FF3D_HALO_ABLATE
  echo -n "swizzle $v: " | tee -a $O/halo_swizzle_ab.txt; timeout 120 python tools/experiments/exp_halo.py 2>&1 | tail -1 | tee -a $O/halo_swizzle_ab.txt
done
unset FF3D_HALO_ABLATE
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  ( cd /tmp && timeout 150 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p -- python $R/tools/experiments/exp_halo.py > $O/pmc_$c.log 2>&1 )
  DB=$(find $O/pmc_$c -name '*_results.db' | head -1)
  [ -n "$DB" ] && python tools/pmc_summary.py $DB halo | grep -v "^#\|dispatches" | head -3 | cut -c1-170 | tee -a $O/pmc_halo_new_swizzle.txt
  find $O/pmc_$c -name '*.db' -delete
done
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['execution'][:16])
PY
}
b() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-strong-probe "$@" > $O/$name.json 2> $O/$name.err; show $O/$name.json; }
FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1_fused --batch 4 --steps 40 --warmup 5
FF3D_BENCH_FORCE_DIST=1 FF3D_LIN_LN=0 FF3D_QKV_FUSED=0 b bench_b4_rccl1_unfused --batch 4 --steps 40 --warmup 5
FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1_fused_2 --batch 4 --steps 40 --warmup 5
b bench_b4_graph_fused --batch 4 --steps 40 --warmup 5
b bench_b1_graph --batch 1 --steps 40 --warmup 5
FF3D_LIN_MIN_ROWS=0 b bench_b1_graph_ownlin --batch 1 --steps 40 --warmup 5
for B in 4 1; do
  ( cd /tmp && FF3D_LIN_MIN_ROWS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_b$B -o r -- python $R/bench.py --batch $B --steps 6 --warmup 3 --graph off --no-cpu-baseline --no-strong-probe > $O/bench_under_rocprof_b$B.json 2> $O/rocprof_b$B.err )
  DB=$(find $O/prof_b$B -name '*_results.db' | head -1)
  python tools/rocprof_last_step.py $DB 70 > $O/bench_b${B}_kernel_stats_last_step.txt 2>&1
  find $O/prof_b$B -name '*.db' -delete
done
head -30 $O/bench_b4_kernel_stats_last_step.txt | cut -c1-150
b bench_b32
