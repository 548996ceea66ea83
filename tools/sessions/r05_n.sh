#!/bin/bash
# round 5 session n: self-attention with the next K / V tile prefetched: parity + kernel time
O=$PWD/gpurun_out/r05_n; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_head_gpu.py -x -q -k "attention or attn or golden or full_size" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -5 $O/tests.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $R/bench.py --graph off --steps 5 --warmup 3 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_b32.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_b32_eager_kernel_stats_last_step.txt 2>&1
rm -rf $O/prof_b32
grep -n "self_attn\|last step" $O/bench_b32_eager_kernel_stats_last_step.txt | cut -c1-170
timeout 400 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_default.json 2> $O/bench_default.err
timeout 400 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --batch 4 --steps 40 > $O/bench_b4.json 2> $O/bench_b4.err
python - <<'PY'
import json
for n in ('default', 'b4'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_n/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['config'].get('single_stream_eager', {}).get('value'))
    except Exception as e:
        print(n, 'no line', e)
PY
