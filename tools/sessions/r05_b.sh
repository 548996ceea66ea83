#!/bin/bash
# round 5 session b: linrows.hip (row-owning linear: split-fp16 + bf16) unit tests, bf16-mode head tests, torch-level graph/sync repro,
# bench default (fused LN rows on) vs FF3D_LIN_ROWS=0, waymo workload
mkdir -p gpurun_out/r05_b
O=gpurun_out/r05_b
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "linear_rows" > $O/tests_rows.log 2>&1; echo "rc=$?" >> $O/tests_rows.log
tail -15 $O/tests_rows.log
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -k "configs0 or engineered" > $O/tests_round5.log 2>&1; echo "rc=$?" >> $O/tests_round5.log
tail -5 $O/tests_round5.log
for v in "inplace device" "alloc device" "none device" "inplace stream" "inplace event" "inplace device head" "alloc device head"; do
  timeout 120 python tools/repro_graph_sync_fault_torch.py $v > "$O/repro_torch_${v// /_}.log" 2>&1; echo "torch variant [$v] rc=$?" >> $O/repro_torch_summary.txt
done
cat $O/repro_torch_summary.txt
timeout 900 python -m pytest tests/test_baseline_configs_gpu.py tests/test_head_gpu.py -x -q -k "bf16" > $O/tests_bf16.log 2>&1; echo "rc=$?" >> $O/tests_bf16.log
tail -8 $O/tests_bf16.log
timeout 900 python -m pytest tests/test_bench_shape_gpu.py -x -q -k "batch32 or batch4" > $O/tests_shape.log 2>&1; echo "rc=$?" >> $O/tests_shape.log
tail -5 $O/tests_shape.log
for mode in ln 0 all; do
  FF3D_LIN_ROWS=$mode timeout 300 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_rows_$mode.json 2> $O/bench_rows_$mode.err
done
timeout 300 python bench.py --workload waymo --no-cpu-baseline --steps 10 > $O/bench_waymo.json 2> $O/bench_waymo.err
python - <<'PY'
import json
for n in ('rows_ln', 'rows_0', 'rows_all', 'waymo'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_b/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if 'linear' in k})
    except Exception as e:
        print(n, 'no line', e)
PY
