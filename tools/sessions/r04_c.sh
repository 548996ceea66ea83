#!/bin/bash
# round 4 session c: pruned library + pipelined bench: whole GPU suite, smoke, default bench line, batch sweep, 1-rank RCCL rehearsals
O=$PWD/gpurun_out/r04_c; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python -m pytest tests/test_small_batch_gpu.py -x -q -m gpu > $O/pytest_small.log 2>&1; echo "small-batch tests rc=$?"; tail -3 $O/pytest_small.log | cut -c1-600
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-200 $O/bench_default.json; tail -3 $O/bench_default.err
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['execution'][:90])
PY
}
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/$name.json 2> $O/$name.err; echo "rc=$?"; show $O/$name.json; tail -2 $O/$name.err | cut -c1-300; }
FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1 --batch 4 --steps 40 --warmup 5
FF3D_BENCH_FORCE_DIST=1 b bench_b32_rccl1 --batch 32
FF3D_BENCH_FORCE_DIST=1 FF3D_BENCH_DIST_MODE=eager b bench_b4_rccl1_eager --batch 4 --steps 40 --warmup 5
b bench_b4 --batch 4 --steps 40 --warmup 5
b bench_b1 --batch 1 --steps 40 --warmup 5
b bench_b8 --batch 8
b bench_b16 --batch 16
b bench_b32_eager --graph off
b bench_b32_slots1 --slots 1
b bench_b32_steps30 --steps 30
b bench_waymo_b8 --workload waymo
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest_all.log | cut -c1-300
