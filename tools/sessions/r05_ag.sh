#!/bin/bash
# round 5 session ag: K-step-tiled weight planes of the halo conv as the default: parity, lc / waymo A/B
O=$PWD/gpurun_out/r05_ag; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_ops_gpu.py tests/test_head_gpu.py -x -q -k "halo or conv or neck or lc or golden" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -n 3 $O/tests.log | cut -c1-200
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b lc_tiled --workload lc --steps 10
FF3D_HALO_W_TILED=0 b lc_rows --workload lc --steps 10
b waymo_tiled --workload waymo --steps 10
FF3D_HALO_W_TILED=0 b waymo_rows --workload waymo --steps 10
python - <<'PY'
import json
for n in ('lc_tiled', 'lc_rows', 'waymo_tiled', 'waymo_rows'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_ag/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))
    except Exception as e:
        print(n, 'no line', e)
PY
