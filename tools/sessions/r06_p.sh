#!/bin/bash
# round 6 session p: gather_first with the gathered rows in two column groups (dual linear: half the projection's K) - tests + A/B; lc A/B re-run
O=$PWD/gpurun_out/r06_p; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_round6_gpu.py -x -q -k "gather_rows or gather_first" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -n "Error\|assert \|passed\|failed" $O/tests.log | head
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b gf_a --value-mode gather_first; b pf_a; b gf_b --value-mode gather_first; b pf_b
b lc_mfma_a --workload lc --steps 10; FF3D_LOCATT_MFMA=0 b lc_scalar_a --workload lc --steps 10; b lc_mfma_b --workload lc --steps 10; FF3D_LOCATT_MFMA=0 b lc_scalar_b --workload lc --steps 10
python - <<'PY'
import json
for n in ('gf_a', 'pf_a', 'gf_b', 'pf_b', 'lc_mfma_a', 'lc_scalar_a', 'lc_mfma_b', 'lc_scalar_b'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r06_p/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['roofline']['avg_launch_ms'], {k: v for k, v in d.get('roofline_dense', {}).get('dense_launches_ms', {}).items() if '1056' in k or '2080' in k})
    except Exception as e:
        print(n, 'no line', e, open(f'gpurun_out/r06_p/bench_{n}.err').read()[-600:])
PY
