#!/bin/bash
# round 5 session h: locatt2 with LDS-DMA staging and 4-row tiles; pair output on the weight-stationary GEMM; lc / default benches
O=$PWD/gpurun_out/r05_h; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_head_gpu.py tests/test_round5_gpu.py -x -q -k "locatt or local_context or neck or weight_stationary or nhwc_pair" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -6 $O/tests.log
timeout 900 python -m pytest tests/test_baseline_configs_gpu.py -x -q -k "config2 or lc_chain" > $O/tests_lc.log 2>&1; echo "rc=$?" >> $O/tests_lc.log
tail -3 $O/tests_lc.log
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b lc --workload lc --steps 10
FF3D_LOCATT_TY=8 b lc_ty8 --workload lc --steps 10
FF3D_GEMM_WS_PAIR=0 b lc_ws_pair_off --workload lc --steps 10
b default
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_lc -o r -- python $R/bench.py --graph off --workload lc --steps 4 --warmup 2 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_lc.json 2> $O/rocprof_lc.err )
DB=$(find $O/prof_lc -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_lc_kernel_stats_last_step.txt 2>&1
rm -rf $O/prof_lc
head -16 $O/bench_lc_kernel_stats_last_step.txt | cut -c1-170
python - <<'PY'
import json
for n in ('lc', 'lc_ty8', 'lc_ws_pair_off', 'default'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_h/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['config'].get('single_stream_eager', {}).get('value'))
    except Exception as e:
        print(n, 'no line', e)
PY
