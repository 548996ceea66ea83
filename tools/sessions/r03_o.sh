#!/bin/bash
# round 3 session o: the shipped build - whole GPU suite, default bench line, rocprofv3 kernel stats of the same command, PMC traffic
# passes, the small-batch / collective / other-workload bench lines, training step
O=$PWD/gpurun_out/r03_o; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest_all.log | cut -c1-300
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-200 $O/bench_default.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-strong-probe > $O/bench_under_rocprof_b32.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_b32_kernel_stats_last_step.txt 2>&1
python tools/rocprof_summary.py $DB 40 > $O/bench_b32_kernel_stats_all.txt 2>&1
find $O/prof_b32 -name '*.db' -delete
head -12 $O/bench_b32_kernel_stats_last_step.txt | cut -c1-150
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_$C -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-probe > $O/pmc_$C.json 2> $O/pmc_$C.err )
  python tools/pmc_summary.py $(find $O/pmc_$C -name '*_results.db' | head -1) msda_fwd conv3x3_halo splitmm split_nchw bev_flatten roi_grid linear > $O/pmc_$C.txt 2>&1
  find $O/pmc_$C -name '*.db' -delete
  grep -i "msda\|halo" $O/pmc_$C.txt | head -4 | cut -c1-170
done
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['execution'][:16])
PY
}
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-strong-probe "$@" > $O/$name.json 2> $O/$name.err; show $O/$name.json; }
FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1 --batch 4 --steps 40 --warmup 5
b bench_b4_graph --batch 4 --steps 40 --warmup 5
b bench_b4_eager --batch 4 --steps 40 --warmup 5 --graph off
b bench_b1_graph --batch 1 --steps 40 --warmup 5
b bench_waymo_b8 --workload waymo
b bench_lc_b8 --workload lc
timeout 600 python tools/bench_train_step.py > $O/train_step.json 2> $O/train_step.err; tail -1 $O/train_step.json | cut -c1-300
C=256 timeout 600 python tools/bench_train_step.py 4 256 > $O/train_step_c256.json 2> $O/train_step_c256.err; tail -1 $O/train_step_c256.json | cut -c1-300
