#!/bin/bash
# round 4 session i: final tree - smoke, whole GPU suite, default bench (20 steps), NMS micro-benchmark of the shipped kernel
O=$PWD/gpurun_out/r04_i; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest_all.log | cut -c1-300
for w in 1 0; do echo "FF3D_NMS_WIDE=$w:" | tee -a $O/nms_wide_ab.txt; FF3D_NMS_WIDE=$w timeout 120 python tools/experiments/exp_nms.py 2>&1 | grep "B=" | tee -a $O/nms_wide_ab.txt; done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_i/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], d['config']['execution'][:60], 'configs3', d['configs3_strong'].get('value'), d['configs3_strong'].get('projected_speedup_8_vs_1'), {k:(v.get('value'),v.get('error')) for k,v in d['other_workloads'].items()})
print('roofline', d['roofline']['frac'], d['roofline']['frac_counter'], 'dense', d['roofline_dense']['frac'], 'cpu', d['cpu_baseline']['value'])
PY
