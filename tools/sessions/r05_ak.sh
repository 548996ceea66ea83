#!/bin/bash
# round 5 session ak: flatten frames-per-block rule (>= 8 192 blocks) on 4 and 32 frames
O=$PWD/gpurun_out/r05_ak; mkdir -p $O
export TMPDIR=/tmp
b() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b b4 --batch 4 --steps 40
b b32
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "flatten" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -n 2 $O/tests.log
python - <<'PY'
import json
for n in ('b4', 'b32'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_ak/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))
    except Exception as e:
        print(n, 'no line', e)
PY
