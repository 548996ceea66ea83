#!/bin/bash
# round 5 session c: one-plane bf16 GEMMs (value_proj weight-stationary, roi_mlp.0 split-K), bf16 flatten, rows box_update; the bf16-mode
# head tests on the own kernels; graph/sync repro with a large graph; waymo bench (slots auto -> 2) + profile; default bench
mkdir -p gpurun_out/r05_c
O=gpurun_out/r05_c
hipcc --offload-arch=gfx950 -O2 tools/repro_graph_sync_fault.hip -o /tmp/repro_graph > $O/repro_build.log 2>&1
timeout 120 /tmp/repro_graph many > $O/repro_many.log 2>&1; echo "variant many rc=$?" > $O/repro_summary.txt; tail -2 $O/repro_many.log >> $O/repro_summary.txt
cat $O/repro_summary.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "linear_rows_bf16 or gemm_bf16 or flatten_bf16 or box_update_rows" > $O/tests_ops.log 2>&1; echo "rc=$?" >> $O/tests_ops.log
tail -12 $O/tests_ops.log
timeout 1200 python -m pytest tests/test_baseline_configs_gpu.py tests/test_head_gpu.py -x -q -k "bf16 or waymo" > $O/tests_bf16.log 2>&1; echo "rc=$?" >> $O/tests_bf16.log
tail -8 $O/tests_bf16.log
timeout 1200 python -m pytest tests/test_head_gpu.py tests/test_bench_shape_gpu.py -x -q -k "golden or full_size or batch4" > $O/tests_head.log 2>&1; echo "rc=$?" >> $O/tests_head.log
tail -5 $O/tests_head.log
timeout 600 python -m pytest tests/test_small_batch_gpu.py -x -q > $O/tests_small.log 2>&1; echo "rc=$?" >> $O/tests_small.log
tail -3 $O/tests_small.log
timeout 400 python bench.py --workload waymo --no-cpu-baseline --steps 10 > $O/bench_waymo.json 2> $O/bench_waymo.err
FF3D_GEMM_WS_BF16_NJ=2 timeout 400 python bench.py --workload waymo --no-cpu-baseline --steps 10 > $O/bench_waymo_nj2.json 2> $O/bench_waymo_nj2.err
timeout 400 python bench.py --workload waymo --no-cpu-baseline --steps 10 --slots 1 > $O/bench_waymo_slots1.json 2> $O/bench_waymo_slots1.err
timeout 300 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_default.json 2> $O/bench_default.err
R=$(pwd); export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_waymo -o r -- python $R/bench.py --workload waymo --graph off --steps 4 --warmup 2 --no-cpu-baseline --no-strong-probe --no-other-workloads > $R/$O/bench_under_rocprof_waymo.json 2> $R/$O/rocprof_waymo.err )
DB=$(find $O/prof_waymo -name "*_results.db" | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/waymo_kernel_stats_last_step.txt 2>&1
head -45 $O/waymo_kernel_stats_last_step.txt | cut -c1-170
rm -rf $O/prof_waymo
python - <<'PY'
import json
for n in ('waymo', 'waymo_nj2', 'waymo_slots1', 'default'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_c/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'], d['config']['execution'][:60])
        print('   ', {k: v for k, v in d['roofline_dense']['dense_launches_ms'].items() if 'gemm' in k or 'rows' in k})
    except Exception as e:
        print(n, 'no line', e)
PY
tail -3 $O/bench_waymo.err
