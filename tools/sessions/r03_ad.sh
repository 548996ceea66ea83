#!/bin/bash
# round 3 session ad: self-attention with the K / V loads two tiles ahead in registers: tests, kernel time at 1 / 4 / 32 frames
O=$PWD/gpurun_out/r03_ad; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or attn" > $O/pytest_attn.log 2>&1; echo "attention tests rc=$?"; tail -2 $O/pytest_attn.log | cut -c1-300
for B in 1 4 32; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_b$B -o r -- python $R/bench.py --batch $B --steps 6 --warmup 3 --graph off --no-cpu-baseline --no-strong-probe > $O/bench_under_rocprof_b$B.json 2> $O/rocprof_b$B.err )
  DB=$(find $O/prof_b$B -name '*_results.db' | head -1)
  python tools/rocprof_last_step.py $DB 70 > $O/bench_b${B}_kernel_stats_last_step.txt 2>&1
  find $O/prof_b$B -name '*.db' -delete
  echo "B=$B"; grep -i "self_attn\|last step" $O/bench_b${B}_kernel_stats_last_step.txt | cut -c1-150
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
b bench_b1_graph --batch 1 --steps 40 --warmup 5
b bench_b32
