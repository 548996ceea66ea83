#!/bin/bash
# round 3 session m: hand-scheduled fragment pipelines (halo conv: inline-asm reads / waits / AGPR MFMAs, 3-stage weight ring; ws GEMM:
# pinned 2-ahead fragment prefetch, 192-column blocks), small-M linear kernel with the deep weight ring: tests, A/B timings, benches
O=$PWD/gpurun_out/r03_m; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "linear or conv or halo or gemm or split" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/pytest_ops.log | cut -c1-400
for v in 1 0 1 0; do
  echo -n "FF3D_HALO_PIPE=$v: " | tee -a $O/halo_pipe_ab.txt; FF3D_HALO_PIPE=$v timeout 120 python tools/experiments/exp_halo.py 2>&1 | tail -1 | tee -a $O/halo_pipe_ab.txt
done
for v in 3 2 3 2; do
  echo -n "FF3D_GEMM_WS_NJ=$v: " | tee -a $O/ws_ab.txt; FF3D_GEMM_WS_NJ=$v timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | tail -1 | tee -a $O/ws_ab.txt
done
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['execution'][:16])
PY
}
b() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-strong-probe "$@" > $O/$name.json 2> $O/$name.err; show $O/$name.json; }
b bench_b32
FF3D_HALO_PIPE=0 b bench_b32_halo_old
FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1 --batch 4 --steps 40 --warmup 5
FF3D_BENCH_FORCE_DIST=1 FF3D_LIN_LN=0 b bench_b4_rccl1_noln --batch 4 --steps 40 --warmup 5
FF3D_BENCH_FORCE_DIST=1 FF3D_LIN_SMALL=0 FF3D_LIN_LN=0 FF3D_QKV_FUSED=0 b bench_b4_rccl1_r03l_config --batch 4 --steps 40 --warmup 5
b bench_b4_graph --batch 4 --steps 40 --warmup 5
b bench_b1_graph --batch 1 --steps 40 --warmup 5
FF3D_LIN_MIN_ROWS=0 b bench_b1_graph_ownlin --batch 1 --steps 40 --warmup 5
FF3D_LIN_MIN_ROWS=0 FF3D_LIN_LN=0 b bench_b1_graph_ownlin_noln --batch 1 --steps 40 --warmup 5
for B in 4 1; do
  ( cd /tmp && FF3D_LIN_MIN_ROWS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_b$B -o r -- python $R/bench.py --batch $B --steps 6 --warmup 3 --graph off --no-cpu-baseline --no-strong-probe > $O/bench_under_rocprof_b$B.json 2> $O/rocprof_b$B.err )
  DB=$(find $O/prof_b$B -name '*_results.db' | head -1)
  python tools/rocprof_last_step.py $DB 70 > $O/bench_b${B}_kernel_stats_last_step.txt 2>&1
  find $O/prof_b$B -name '*.db' -delete
done
head -14 $O/bench_b4_kernel_stats_last_step.txt | cut -c1-150
head -14 $O/bench_b1_kernel_stats_last_step.txt | cut -c1-150
timeout 1200 python -m pytest tests/test_head_gpu.py tests/test_bench_shape_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu > $O/pytest_head.log 2>&1; echo "head rc=$?"; tail -3 $O/pytest_head.log | cut -c1-300
