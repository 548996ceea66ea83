#!/bin/bash
# round 5 session e: lc as one graph (fixed), frame-fastest flatten grid A/B, paired 16-byte bf16 stores, waymo / default / lc benches
mkdir -p gpurun_out/r05_e
O=gpurun_out/r05_e
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -k "neck_and_head" > $O/tests_lc.log 2>&1; echo "rc=$?" >> $O/tests_lc.log
tail -6 $O/tests_lc.log
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm_bf16 or flatten or roi" > $O/tests_ops.log 2>&1; echo "rc=$?" >> $O/tests_ops.log
tail -4 $O/tests_ops.log
timeout 900 python -m pytest tests/test_baseline_configs_gpu.py tests/test_head_gpu.py -x -q -k "bf16 or waymo or full_size" > $O/tests_head.log 2>&1; echo "rc=$?" >> $O/tests_head.log
tail -4 $O/tests_head.log
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b lc --workload lc --steps 10
b lc_slots1 --workload lc --steps 10 --slots 1
b waymo --workload waymo --steps 10
FF3D_FLATTEN_ORDER=frame-slowest b waymo_flatten_old --workload waymo --steps 10
b default
FF3D_FLATTEN_ORDER=frame-slowest b default_flatten_old
b default2
python - <<'PY'
import json
for n in ('lc', 'lc_slots1', 'waymo', 'waymo_flatten_old', 'default', 'default_flatten_old', 'default2'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_e/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['config']['execution'][:60], {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if 'bf16' in k})
    except Exception as e:
        print(n, 'no line', e)
PY
tail -3 $O/bench_lc.err
