#!/bin/bash
# round 6 session y: rocprofv3 kernel tables of the round's tree - the default command (graph replay) and the eager 32-frame step; lc and waymo eager steps
O=$PWD/gpurun_out/r06_y; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() { name=$1; shift; ( cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o r -- python $R/bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions "$@" > $O/bench_under_rocprof_$name.json 2> $O/rocprof_$name.err ); DB=$(find $O/prof_$name -name '*_results.db' | head -1); python tools/rocprof_last_step.py $DB 60 > $O/bench_${name}_kernel_stats_last_step.txt 2>&1; python tools/rocprof_summary.py $DB 30 > $O/bench_${name}_kernel_stats.txt 2>&1; rm -rf $O/prof_$name; head -4 $O/bench_${name}_kernel_stats_last_step.txt | cut -c1-150; }
prof default
prof b32_eager --graph off --steps 5 --warmup 3
prof lc_eager --workload lc --graph off --steps 5 --warmup 3
prof waymo_eager --workload waymo --graph off --steps 5 --warmup 3
grep -h "msda_fwd_kernel" $O/bench_default_kernel_stats.txt $O/bench_b32_eager_kernel_stats_last_step.txt | cut -c1-150
python - <<'PY'
import json
for n in ('default', 'b32_eager', 'lc_eager', 'waymo_eager'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r06_y/bench_under_rocprof_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
    except Exception as e:
        print(n, 'no line', e)
PY
