#!/bin/bash
# round 5 session af: halo conv with K-step-tiled weight planes (FF3D_HALO_W_TILED=1) against row-major ones: A/B on the 32-frame step
O=$PWD/gpurun_out/r05_af; mkdir -p $O
export TMPDIR=/tmp
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
FF3D_HALO_W_TILED=1 b tiled
b rows
FF3D_HALO_W_TILED=1 b tiled2
b rows2
python - <<'PY'
import json
for n in ('tiled', 'rows', 'tiled2', 'rows2'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_af/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if 's1 180' in k})
    except Exception as e:
        print(n, 'no line', e)
PY
FF3D_HALO_W_TILED=1 timeout 600 python -m pytest tests/test_bench_shape_gpu.py -x -q > $O/tests_tiled.log 2>&1; echo "rc=$?" >> $O/tests_tiled.log
tail -n 3 $O/tests_tiled.log | cut -c1-200
