#!/bin/bash
# round 4 session t: last check of the tree after the final small commits - whole GPU suite with -x (as the driver runs it), smoke, default bench
O=$PWD/gpurun_out/r04_t; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest_all.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
/usr/bin/time -v timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; grep "Elapsed (wall clock)" $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_t/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], 'configs3', d['configs3_strong'].get('value'), d['configs3_strong'].get('projected_speedup_8_vs_1'), {k:(v.get('value'),v.get('error')) for k,v in d['other_workloads'].items()})
print('roofline', d['roofline']['frac'], d['roofline']['frac_counter'], d['roofline']['timed_in'][:40], 'dense', d['roofline_dense']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['protocol'])
PY
