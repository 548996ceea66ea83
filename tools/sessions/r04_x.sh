#!/bin/bash
# round 4 session x: one flatten pass for both decoder stages in bf16 / vendor mode - head goldens, waymo parity, A/B
O=$PWD/gpurun_out/r04_x; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_head_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu -k "head or waymo or bf16" > $O/pytest_sel.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest_sel.log | cut -c1-400
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'])
PY
}
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/$name.json 2> $O/$name.err; echo "rc=$?"; show $O/$name.json; }
b bench_waymo_multi --workload waymo
FF3D_FLATTEN_MULTI_F32=0 b bench_waymo_per_stage --workload waymo
b bench_waymo_multi_2 --workload waymo
