#!/bin/bash
# round 4 session j: small-batch + bench CLI tests on the final tree (own linear kernels while replays overlap), batch sweep
O=$PWD/gpurun_out/r04_j; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_small_batch_gpu.py tests/test_bench_cli_gpu.py -q -m gpu > $O/pytest_sel.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest_sel.log | cut -c1-400
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['execution'][:70])
PY
}
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/$name.json 2> $O/$name.err; echo "rc=$?"; show $O/$name.json; }
b bench_b1 --batch 1 --steps 60 --warmup 5
b bench_b2 --batch 2 --steps 60 --warmup 5
b bench_b4 --batch 4 --steps 40 --warmup 5
FF3D_BENCH_FORCE_DIST=1 b bench_b4_rccl1 --batch 4 --steps 40 --warmup 5
