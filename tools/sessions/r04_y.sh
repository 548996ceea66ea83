#!/bin/bash
# round 4 session y: the last tree - whole GPU suite (-x), smoke, default bench
O=$PWD/gpurun_out/r04_y; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest_all.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
T0=$(date +%s.%N)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
T1=$(date +%s.%N); echo "bench wall s: $(python -c "print(round($T1-$T0,1))")" | tee $O/bench_wall.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_y/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], 'configs3', d['configs3_strong'].get('value'), d['configs3_strong'].get('projected_speedup_8_vs_1'), {k:(v.get('value'),v.get('error')) for k,v in d['other_workloads'].items()})
print('roofline', d['roofline']['frac'], d['roofline']['frac_counter'], 'dense', d['roofline_dense']['frac'], 'cpu', d['cpu_baseline']['value'])
PY
