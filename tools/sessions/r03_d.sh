#!/bin/bash
# round 3 session d: linear kernel v2, updated BASELINE config tests, whole suite, benches incl. graph + 1-rank RCCL
O=$PWD/gpurun_out/r03_d; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "linear_f16x3" > $O/pytest_linear.log 2>&1; echo "linear rc=$?"; tail -4 $O/pytest_linear.log | cut -c1-300
FF3D_PARITY_STATS=$O/stats timeout 1200 python -m pytest tests/test_baseline_configs_gpu.py -q -m gpu > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_new.log | cut -c1-400 | head -20
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_baseline_configs_gpu.py > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -5 $O/pytest_all.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline > $O/bench_b32.json 2> $O/bench_b32.err; echo "b32 rc=$?"; python - <<PY
import json
d=json.loads(open('$O/bench_b32.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(d.get('configs3_strong')); print(d['roofline_dense']['dense_launches_ms'])
PY
timeout 300 python bench.py --batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-strong-probe > $O/bench_b4.json 2> $O/bench_b4.err; python - <<PY
import json
d=json.loads(open('$O/bench_b4.json').read().strip().splitlines()[-1])
print('b4', d['value'], d['ms_per_step'], d['config']['execution']); print(d['roofline_dense']['dense_launches_ms'])
PY
timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-strong-probe > $O/bench_b1.json 2> $O/bench_b1.err; cut -c70-130 $O/bench_b1.json
FF3D_BENCH_FORCE_DIST=1 timeout 300 python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline --no-strong-probe > $O/bench_b4_rccl1.json 2> $O/bench_b4_rccl1.err; echo "rccl1 rc=$?"; cut -c70-130 $O/bench_b4_rccl1.json; tail -2 $O/bench_b4_rccl1.err
