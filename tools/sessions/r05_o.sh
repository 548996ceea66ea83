#!/bin/bash
# round 5 session o: tail conv with the A fragments shared by the three dy taps; BatchNorm folds kept per weight version (lc dispatches)
O=$PWD/gpurun_out/r05_o; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_head_gpu.py tests/test_bench_shape_gpu.py tests/test_round5_gpu.py -x -q -k "tail or small or heatmap or golden or full_size or neck or lc or pipelined" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -5 $O/tests.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $R/bench.py --graph off --steps 5 --warmup 3 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_b32.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_b32_eager_kernel_stats_last_step.txt 2>&1
rm -rf $O/prof_b32
grep -n "conv3x3_small\|last step" $O/bench_b32_eager_kernel_stats_last_step.txt | cut -c1-170
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_lc -o r -- python $R/bench.py --graph off --workload lc --steps 4 --warmup 2 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_lc.json 2> $O/rocprof_lc.err )
DB=$(find $O/prof_lc -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 70 > $O/bench_lc_kernel_stats_last_step.txt 2>&1
rm -rf $O/prof_lc
head -3 $O/bench_lc_kernel_stats_last_step.txt | cut -c1-170
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b default
b lc --workload lc --steps 12
python - <<'PY'
import json
for n in ('default', 'lc'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_o/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['config'].get('single_stream_eager', {}).get('value'))
    except Exception as e:
        print(n, 'no line', e)
PY
