#!/bin/bash
# round 6, third session: validation of the final tree (smoke, whole GPU suite, default bench line)
O=$PWD/gpurun_out/r06_v9; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
( time timeout 3300 python -m pytest tests -x -q -m gpu ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -8 $O/tests.log | cut -c1-300
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06_v9/bench_default.json') if l.startswith('{')][-1])
print('value', d['value'], d['ms_per_step'], d['verified'])
print('fresh', json.dumps(d['config'].get('fresh_inputs'))[:300])
print('gather_first', json.dumps(d['config'].get('value_mode_gather_first'))[:300])
print('latency', json.dumps(d.get('latency_b1_ms'))[:300])
print('other', {k: (v.get('value'), v.get('error')) for k, v in d.get('other_workloads', {}).items()})
print('cpu', d.get('cpu_baseline', {}).get('value'))
print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
