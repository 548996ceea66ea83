#!/bin/bash
# round 5 session i: batches in flight for the lc and waymo workloads (2 / 3 / 4 slots)
O=$PWD/gpurun_out/r05_i; mkdir -p $O
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b lc_s2 --workload lc --steps 12 --slots 2
b lc_s3 --workload lc --steps 12 --slots 3
b lc_s4 --workload lc --steps 12 --slots 4
b waymo_s2 --workload waymo --steps 10 --slots 2
b waymo_s3 --workload waymo --steps 12 --slots 3
b l_b4_s4 --batch 4 --steps 40 --warmup 5
b l_b1 --batch 1 --steps 60 --warmup 5
python - <<'PY'
import json
for n in ('lc_s2', 'lc_s3', 'lc_s4', 'waymo_s2', 'waymo_s3', 'l_b4_s4', 'l_b1'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_i/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['config']['execution'][:50], d['config'].get('single_stream_eager', {}).get('value'))
    except Exception as e:
        print(n, 'no line', e)
PY
tail -2 $O/bench_lc_s4.err
