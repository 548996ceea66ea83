#!/bin/bash
# round 3 session x: the final tree - whole GPU suite, smoke, default bench, the other workloads and batch sizes
O=$PWD/gpurun_out/r03_x; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest_all.log | cut -c1-300
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-200 $O/bench_default.json
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
b bench_b8 --batch 8
b bench_b16 --batch 16
b bench_b64 --batch 64 --steps 6
b bench_waymo_b8 --workload waymo
b bench_lc_b8 --workload lc
