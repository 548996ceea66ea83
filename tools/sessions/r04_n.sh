#!/bin/bash
# round 4 session n: the final tree - smoke, whole GPU suite, default bench, 4-frame lines
O=$PWD/gpurun_out/r04_n; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest_all.log | cut -c1-300
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_n/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], d['config']['execution'][:60], 'configs3', d['configs3_strong'].get('value'), d['configs3_strong'].get('projected_speedup_8_vs_1'), {k:(v.get('value'),v.get('error')) for k,v in d['other_workloads'].items()})
print('roofline', d['roofline']['frac'], d['roofline']['frac_counter'], 'dense', d['roofline_dense']['frac'], 'cpu', d['cpu_baseline']['value'])
PY
timeout 400 python bench.py --batch 4 --steps 40 --warmup 5 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_b4.json 2> $O/bench_b4.err; python -c "
import json; d=json.loads(open('gpurun_out/r04_n/bench_b4.json').read().strip().splitlines()[-1]); print('b4', d['value'], d['ms_per_step'])"
