#!/bin/bash
# round 6 session g: value GEMM with 256 columns per block (splitmm_ws1_kernel) - unit test, microbench A/B, step A/B
O=$PWD/gpurun_out/r06_g; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_round6_gpu.py tests/test_ops_gpu.py -x -q -k "gemm" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -5 $O/tests.log | cut -c1-250
for v in 1 0 1 0; do FF3D_GEMM_WS1=$v timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | sed "s/^/WS1=$v /" >> $O/microbench.txt; done
for n in 256 512; do for v in 1 0; do N=$n FF3D_GEMM_WS1=$v timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | sed "s/^/WS1=$v /" >> $O/microbench.txt; done; done
for b in 4 1; do for v in 1 0; do B=$b FF3D_GEMM_WS1=$v timeout 200 python tools/experiments/exp_valuegemm.py 2>&1 | sed "s/^/WS1=$v /" >> $O/microbench.txt; done; done
cat $O/microbench.txt
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b ws1_a; FF3D_GEMM_WS1=0 b ws0_a; b ws1_b; FF3D_GEMM_WS1=0 b ws0_b
python - <<'PY'
import json
for n in ('ws1_a', 'ws0_a', 'ws1_b', 'ws0_b'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r06_g/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if 'gemm 1360800' in k})
    except Exception as e:
        print(n, 'no line', e)
PY
