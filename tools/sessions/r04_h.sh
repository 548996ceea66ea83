#!/bin/bash
# round 4 session h: wide NMS kernel (LDS histogram, 2-D thread mapping) - tests + A/B; bench.py CLI tests; waymo fp32-class with two
# batches in flight (is the two-slot hang the vendor bf16 GEMMs'?); default bench
O=$PWD/gpurun_out/r04_h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_bench_cli_gpu.py tests/test_head_gpu.py -x -q -m gpu -k "nms or topk or bench or head_forward or golden" > $O/pytest_sel.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest_sel.log | cut -c1-400
for w in 1 0 1 0; do echo "FF3D_NMS_WIDE=$w:" | tee -a $O/nms_wide_ab.txt; FF3D_NMS_WIDE=$w timeout 120 python tools/experiments/exp_nms.py 2>&1 | grep "B=" | tee -a $O/nms_wide_ab.txt; done
PYTHONFAULTHANDLER=1 timeout -s ABRT 150 python bench.py --workload waymo --gemm-dtype f32 --slots 2 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_waymo_f32_slots2.json 2> $O/bench_waymo_f32_slots2.err; echo "waymo f32 2 slots rc=$?"; cut -c1-160 $O/bench_waymo_f32_slots2.json
timeout 200 python bench.py --workload waymo --gemm-dtype f32 --slots 1 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_waymo_f32_slots1.json 2> $O/bench_waymo_f32_slots1.err; echo "waymo f32 1 slot rc=$?"; cut -c1-160 $O/bench_waymo_f32_slots1.json
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_h/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['execution'][:60], 'configs3', d['configs3_strong'].get('value'), d['configs3_strong'].get('projected_speedup_8_vs_1'), {k:(v.get('value'),v.get('error')) for k,v in d['other_workloads'].items()})
PY
