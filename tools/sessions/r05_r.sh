#!/bin/bash
# round 5 session r: the fused feed-forward kernel (ffnrows.hip): parity, A/B against the two-launch form, kernel time
O=$PWD/gpurun_out/r05_r; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "ffn_rows" > $O/tests_ffn.log 2>&1; echo "rc=$?" >> $O/tests_ffn.log
tail -25 $O/tests_ffn.log | cut -c1-220
timeout 1200 python -m pytest tests/test_bench_shape_gpu.py tests/test_round5_gpu.py tests/test_head_gpu.py -x -q -k "full_size or pipelined or bench_shape or golden" > $O/tests_head.log 2>&1; echo "rc=$?" >> $O/tests_head.log
tail -5 $O/tests_head.log | cut -c1-220
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $R/bench.py --graph off --steps 5 --warmup 3 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_b32.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_b32_eager_kernel_stats_last_step.txt 2>&1
rm -rf $O/prof_b32
grep -n "ffn\|linear\|last step" $O/bench_b32_eager_kernel_stats_last_step.txt | cut -c1-170
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b fused
FF3D_FFN_FUSED=0 b two_launch
b fused2
FF3D_FFN_FUSED=0 b two_launch2
python - <<'PY'
import json
for n in ('fused', 'two_launch', 'fused2', 'two_launch2'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_r/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['config'].get('single_stream_eager', {}).get('value'))
    except Exception as e:
        print(n, 'no line', e)
PY
