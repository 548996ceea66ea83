#!/bin/bash
# round 5 session k: residual pairs on the weight-stationary GEMM; lc bench + kernel table; default bench
O=$PWD/gpurun_out/r05_k; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_head_gpu.py tests/test_baseline_configs_gpu.py tests/test_round5_gpu.py tests/test_ops_gpu.py -x -q -k "neck or config2 or lc_chain or neck_and_head or weight_stationary or nhwc_pair or value_gemm or gemm" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -8 $O/tests.log
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b lc --workload lc --steps 12
b default
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_lc -o r -- python $R/bench.py --graph off --workload lc --steps 4 --warmup 2 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_lc.json 2> $O/rocprof_lc.err )
DB=$(find $O/prof_lc -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_lc_kernel_stats_last_step.txt 2>&1
rm -rf $O/prof_lc
head -14 $O/bench_lc_kernel_stats_last_step.txt | cut -c1-170
python - <<'PY'
import json
for n in ('lc', 'default'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_k/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['config'].get('single_stream_eager', {}).get('value'))
    except Exception as e:
        print(n, 'no line', e)
PY
