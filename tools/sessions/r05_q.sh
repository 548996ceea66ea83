#!/bin/bash
# round 5 session q: evidence on the committed tree - smoke(), the whole GPU suite, the full default bench line (strong probe, other
# workloads, cpu_baseline), the same command under rocprofv3 --kernel-trace --stats (the roofline kernel's average duration)
O=$PWD/gpurun_out/r05_q; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
tail -2 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu_suite.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_suite.txt
tail -4 $O/pytest_gpu_suite.txt
timeout 900 python bench.py > $O/bench_default_full.json 2> $O/bench_default_full.err; echo "bench rc=$?"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python $R/bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof.json 2> $O/rocprof.err )
DB=$(find $O/prof -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_default_kernel_stats_last_step.txt 2>&1
STATS=$(find $O/prof -name '*kernel_stats.csv' | head -1)
[ -n "$STATS" ] && head -40 "$STATS" > $O/bench_default_kernel_stats.csv
rm -rf $O/prof
grep -n "msda\|last step" $O/bench_default_kernel_stats_last_step.txt | cut -c1-170
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r05_q/bench_default_full.json') if l.startswith('{')][-1])
print('value', d['value'], d['ms_per_step'], d['verified'])
print('roofline', d['roofline'])
print('cpu', d.get('cpu_baseline'))
print('strong', d.get('configs3_strong'))
print('other', d.get('other_workloads'))
PY
