#!/bin/bash
# round 6 session q: stride-2 pyramid convs with swapped operands - tests, step A/B, kernel table
O=$PWD/gpurun_out/r06_q; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_ops_gpu.py tests/test_head_gpu.py -x -q -k "stride2 or conv3x3 or full_size or golden" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -n "Error\|assert \|passed\|failed" $O/tests.log | head
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b swap_a; FF3D_CONV_S2_SWAP=0 b noswap_a; b swap_b; FF3D_CONV_S2_SWAP=0 b noswap_b
python - <<'PY'
import json
for n in ('swap_a', 'noswap_a', 'swap_b', 'noswap_b'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r06_q/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if ' s2 ' in k})
    except Exception as e:
        print(n, 'no line', e, open(f'gpurun_out/r06_q/bench_{n}.err').read()[-500:])
PY
