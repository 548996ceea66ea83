#!/bin/bash
# round 5 session ae: is there a fixed cost inside the 20-step timed region?  K = 20 / 40 / 80 / 200 with 2 and 16 warm replays per slot
O=$PWD/gpurun_out/r05_ae; mkdir -p $O
export TMPDIR=/tmp
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --warmup 5 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b k20 --steps 20
b k40 --steps 40
b k80 --steps 80
b k200 --steps 200
FF3D_BENCH_WARM_REPLAYS=16 b k20_w16 --steps 20
FF3D_BENCH_WARM_REPLAYS=8 b k20_w8 --steps 20
FF3D_BENCH_WARM_REPLAYS=16 b k40_w16 --steps 40
b k20b --steps 20
FF3D_BENCH_WARM_REPLAYS=16 b k20_w16b --steps 20
python - <<'PY'
import json
for n in ('k20', 'k40', 'k80', 'k200', 'k20_w16', 'k20_w8', 'k40_w16', 'k20b', 'k20_w16b'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_ae/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['steps'], round(d['ms_per_step'] * d['steps'], 1), d['verified'].get('bit_identical'))
    except Exception as e:
        print(n, 'no line', e)
PY
