#!/bin/bash
# round 5 session x: transposing split with the channel tiles fastest (FF3D_SPLIT_ORDER=pixel: rounds 1-4): parity + A/B on the three workloads
O=$PWD/gpurun_out/r05_x; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_head_gpu.py -x -q -k "split or golden or conv" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log | cut -c1-200
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b waymo_c --workload waymo --steps 10
FF3D_SPLIT_ORDER=pixel b waymo_p --workload waymo --steps 10
b l_c
FF3D_SPLIT_ORDER=pixel b l_p
b lc_c --workload lc --steps 10
FF3D_SPLIT_ORDER=pixel b lc_p --workload lc --steps 10
b waymo_c2 --workload waymo --steps 10
FF3D_SPLIT_ORDER=pixel b waymo_p2 --workload waymo --steps 10
python - <<'PY'
import json
for n in ('waymo_c', 'waymo_p', 'waymo_c2', 'waymo_p2', 'l_c', 'l_p', 'lc_c', 'lc_p'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_x/bench_{n}.json') if l.startswith('{')][-1])
        dl = d.get('roofline_dense', {}).get('dense_launches_ms', {})
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))
    except Exception as e:
        print(n, 'no line', e)
PY
