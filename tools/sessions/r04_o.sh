#!/bin/bash
# round 4 session o: 'bevfusion' neck block with its 1x1 convs as split-fp16 GEMMs on NHWC pairs - parity tests, lc workload A/B
O=$PWD/gpurun_out/r04_o; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_head_gpu.py tests/test_baseline_configs_gpu.py tests/test_training_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "neck or lc_chain or encoder or locatt or local_attention or fused" > $O/pytest_neck.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest_neck.log | cut -c1-400
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]).read().strip().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d.get('top_kernel'))
PY
}
b() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/$name.json 2> $O/$name.err; echo "rc=$?"; show $O/$name.json; }
b bench_lc_pairs --workload lc
FF3D_NECK_PAIR_1X1=0 b bench_lc_vendor --workload lc
b bench_lc_pairs_2 --workload lc
