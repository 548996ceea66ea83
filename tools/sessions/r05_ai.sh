#!/bin/bash
# round 5 session ai: tail conv with chunk-tiled weight planes (default) against row-major (FF3D_TAIL_W_TILED=0): parity + A/B
O=$PWD/gpurun_out/r05_ai; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_ops_gpu.py tests/test_head_gpu.py -x -q -k "tail or small or heatmap or tiled or golden" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -n 3 $O/tests.log | cut -c1-200
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b tiled
FF3D_TAIL_W_TILED=0 b rows
b tiled2
FF3D_TAIL_W_TILED=0 b rows2
python - <<'PY'
import json
for n in ('tiled', 'rows', 'tiled2', 'rows2'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_ai/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))
    except Exception as e:
        print(n, 'no line', e)
PY
