#!/bin/bash
# round 6 session n: value mode gather_first - step A/B at 32 frames and at 4 frames, kernel table of the gather_first step
O=$PWD/gpurun_out/r06_n; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b gf_a --value-mode gather_first; b pf_a; b gf_b --value-mode gather_first; b pf_b
b gf_b4 --value-mode gather_first --batch 4; b pf_b4 --batch 4
python - <<'PY'
import json
for n in ('gf_a', 'pf_a', 'gf_b', 'pf_b', 'gf_b4', 'pf_b4'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r06_n/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'), d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['single_stream_eager']['value'])
    except Exception as e:
        print(n, 'no line', e, open(f'gpurun_out/r06_n/bench_{n}.err').read()[-800:])
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python $R/bench.py --value-mode gather_first --graph off --steps 5 --warmup 3 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions > $O/bench_under_rocprof.json 2> $O/rocprof.err )
DB=$(find $O/prof -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 40 > $O/bench_gather_first_kernel_stats_last_step.txt 2>&1
rm -rf $O/prof
head -24 $O/bench_gather_first_kernel_stats_last_step.txt | cut -c1-160
