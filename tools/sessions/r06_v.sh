#!/bin/bash
# round 6 session v: value mode gather_first on configs[4] (468 x 468, 1000 queries, bf16 decoder GEMMs): bf16 head tests, A/B
O=$PWD/gpurun_out/r06_v; mkdir -p $O
export TMPDIR=/tmp
b() { name=$1; shift; timeout 600 python bench.py --workload waymo --steps 10 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b gf_a --value-mode gather_first; b pf_a; b gf_b --value-mode gather_first; b pf_b
b gf_f32 --value-mode gather_first --gemm-dtype f32; b pf_f32 --gemm-dtype f32
python - <<'PY'
import json
for n in ('gf_a', 'pf_a', 'gf_b', 'pf_b', 'gf_f32', 'pf_f32'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r06_v/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['roofline']['avg_launch_ms'], d['config']['batches_in_flight'])
    except Exception as e:
        print(n, 'no line', e, open(f'gpurun_out/r06_v/bench_{n}.err').read()[-700:])
PY
