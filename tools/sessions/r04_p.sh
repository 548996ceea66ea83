#!/bin/bash
# round 4 session p: the final tree - whole GPU suite, rocprofv3 kernel stats of the eager 32-frame step (final kernels), default bench
O=$PWD/gpurun_out/r04_p; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest_all.log | cut -c1-300
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $R/bench.py --graph off --steps 6 --warmup 3 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_b32_eager.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_b32_eager_kernel_stats_last_step.txt 2>&1
find $O/prof_b32 -name '*.db' -delete
head -16 $O/bench_b32_eager_kernel_stats_last_step.txt | cut -c1-150
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b4 -o r -- python $R/bench.py --graph off --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_b4_eager.json 2> $O/rocprof_b4.err )
DB=$(find $O/prof_b4 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_b4_eager_kernel_stats_last_step.txt 2>&1
find $O/prof_b4 -name '*.db' -delete
head -6 $O/bench_b4_eager_kernel_stats_last_step.txt | cut -c1-150
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_p/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], 'configs3', d['configs3_strong'].get('value'), d['configs3_strong'].get('projected_speedup_8_vs_1'), {k:(v.get('value'),v.get('error')) for k,v in d['other_workloads'].items()})
print('roofline', d['roofline']['frac'], d['roofline']['frac_counter'], 'dense', d['roofline_dense']['frac'], 'cpu', d['cpu_baseline']['value'])
PY
