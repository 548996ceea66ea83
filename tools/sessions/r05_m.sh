#!/bin/bash
# round 5 session m: weight-stationary GEMM with 8 waves x 16 columns (two waves per SIMD) vs 4 x 32: parity + A/B on the default / lc steps
O=$PWD/gpurun_out/r05_m; mkdir -p $O
FF3D_GEMM_WS_WAVES=8 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_bench_shape_gpu.py -x -q -k "weight_stationary or value or gemm_f16x3 or nhwc_pair" > $O/tests_w8.log 2>&1; echo "rc=$?" >> $O/tests_w8.log
tail -5 $O/tests_w8.log
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b default_w4
FF3D_GEMM_WS_WAVES=8 b default_w8
b default_w4b
FF3D_GEMM_WS_WAVES=8 b default_w8b
b lc_w4 --workload lc --steps 12
FF3D_GEMM_WS_WAVES=8 b lc_w8 --workload lc --steps 12
python - <<'PY'
import json
for n in ('default_w4', 'default_w8', 'default_w4b', 'default_w8b', 'lc_w4', 'lc_w8'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_m/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if k.startswith('gemm 1360800') or k.startswith('gemm 259200x256x256')})
    except Exception as e:
        print(n, 'no line', e)
PY
