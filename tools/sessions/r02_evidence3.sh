#!/bin/bash
# round 2 evidence session 3 (final build): default bench (+ CPU baselines), rocprofv3 kernel stats, batch sweep (graph auto)
O=$PWD/gpurun_out/r02_ev3; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench_b32.json 2> $O/bench_b32.err; cut -c1-160 $O/bench_b32.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_b32_under_rocprof.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/b32_kernel_stats_last_step.txt 2>&1
find $O -name '*.db' -delete
for B in 1 2 4 8 16 64; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_b${B}.json 2> $O/bench_b${B}.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r02_ev3/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(os.path.basename(f), d.get('value'), d.get('ms_per_step'), d['config'].get('execution'))
    except Exception as e:
        print(os.path.basename(f), 'FAILED', str(e)[:80])
PY
head -14 $O/b32_kernel_stats_last_step.txt | cut -c1-130
