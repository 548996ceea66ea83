#!/bin/bash
# round 5 session w: scheduling knobs of the pipelined step: 3 slots, staggered stream priorities (same box A/B)
O=$PWD/gpurun_out/r05_w; mkdir -p $O
export TMPDIR=/tmp
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b s2
b s3 --slots 3
FF3D_SLOT_PRIORITY=staggered b s2_prio
b s2b
FF3D_SLOT_PRIORITY=staggered b s2_prio_b
python - <<'PY'
import json
for n in ('s2', 's3', 's2_prio', 's2b', 's2_prio_b'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_w/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['config'].get('single_stream_eager', {}).get('value'))
    except Exception as e:
        print(n, 'no line', e)
PY
