#!/bin/bash
# round 6 session w: gather_first over the un-embedded pyramid + per-layer positional tables - tests, A/B against the first form and the default
O=$PWD/gpurun_out/r06_w; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_round6_gpu.py -x -q -k "gather_rows or gather_first" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -n "Error\|assert \|passed\|failed" $O/tests.log | head
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b tab_a --value-mode gather_first; FF3D_GATHER_FIRST_TABLES=0 b emb_a --value-mode gather_first; b pf_a
b tab_b --value-mode gather_first; FF3D_GATHER_FIRST_TABLES=0 b emb_b --value-mode gather_first; b pf_b
python - <<'PY'
import json
for n in ('tab_a', 'emb_a', 'pf_a', 'tab_b', 'emb_b', 'pf_b'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r06_w/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['roofline']['avg_launch_ms'])
    except Exception as e:
        print(n, 'no line', e, open(f'gpurun_out/r06_w/bench_{n}.err').read()[-700:])
PY
