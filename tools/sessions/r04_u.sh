#!/bin/bash
# round 4 session u: default bench of the final tree with its wall time
O=$PWD/gpurun_out/r04_u; mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s.%N)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
T1=$(date +%s.%N); echo "bench wall s: $(python -c "print(round($T1-$T0,1))")" | tee $O/bench_wall.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_u/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], 'configs3', d['configs3_strong'].get('value'), d['configs3_strong'].get('projected_speedup_8_vs_1'), {k:(v.get('value'),v.get('error')) for k,v in d['other_workloads'].items()})
print('roofline', d['roofline']['frac'], d['roofline']['frac_counter'], d['roofline']['timed_in'][:40], 'dense', d['roofline_dense']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['protocol'])
PY
