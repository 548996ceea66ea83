#!/bin/bash
# round 5 session aj: small batches on the final tree (1 and 4 frames per step)
O=$PWD/gpurun_out/r05_aj; mkdir -p $O
export TMPDIR=/tmp
b() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b b1 --batch 1 --steps 100
b b4 --batch 4 --steps 40
FF3D_FLATTEN_FB=1 b b4_fb1 --batch 4 --steps 40
python - <<'PY'
import json
for n in ('b1', 'b4', 'b4_fb1'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_aj/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['config'].get('batches_in_flight'))
    except Exception as e:
        print(n, 'no line', e)
PY
