#!/bin/bash
# round 4 session f: wide heatmap-NMS kernel - parity tests, A/B against the tile kernel, default bench
O=$PWD/gpurun_out/r04_f; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_head_gpu.py tests/test_training_gpu.py -x -q -m gpu -k "nms or topk or head or heuristic or heatmap" > $O/pytest_nms.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest_nms.log | cut -c1-300
for w in 1 0 1 0; do echo "FF3D_NMS_WIDE=$w:" | tee -a $O/nms_wide_ab.txt; FF3D_NMS_WIDE=$w timeout 120 python tools/experiments/exp_nms.py 2>&1 | grep "B=" | tee -a $O/nms_wide_ab.txt; done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-200 $O/bench_default.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_f/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['execution'], d['configs3_strong'].get('value'), {k:(v.get('value'),v.get('error')) for k,v in d['other_workloads'].items()})
PY
