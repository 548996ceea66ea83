#!/bin/bash
# round 5 session am: lc with the heatmap heads' convs as single launches (tiled weight planes) instead of the grouped launch (row-major planes)
O=$PWD/gpurun_out/r05_am; mkdir -p $O
export TMPDIR=/tmp
b() { name=$1; shift; timeout 200 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
FF3D_HEATMAP_GROUPED=0 b lc_single --workload lc --steps 10
b lc_grouped --workload lc --steps 10
python - <<'PY'
import json
for n in ('lc_single', 'lc_grouped'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_am/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))
    except Exception as e:
        print(n, 'no line', e)
PY
