#!/bin/bash
# round 5 session ah: evidence on the shipped tree - smoke(), the whole GPU suite, the full default bench line, the kernel table of the
# eager 32-frame step, the default command under rocprofv3 (roofline kernel's average duration), a 400-step soak of the pipelined graphs
O=$PWD/gpurun_out/r05_ah; mkdir -p $O
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
rm -rf $O/prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_b32 -o r -- python $R/bench.py --graph off --steps 5 --warmup 3 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_under_rocprof_b32.json 2> $O/rocprof_b32.err )
DB=$(find $O/prof_b32 -name '*_results.db' | head -1)
python tools/rocprof_last_step.py $DB 60 > $O/bench_b32_eager_kernel_stats_last_step.txt 2>&1
rm -rf $O/prof_b32
grep -n "msda\|last step" $O/bench_default_kernel_stats_last_step.txt $O/bench_b32_eager_kernel_stats_last_step.txt | cut -c1-200
timeout 600 python bench.py --steps 400 --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_soak_400.json 2> $O/bench_soak_400.err; echo "soak rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r05_ah/bench_default_full.json') if l.startswith('{')][-1])
print('value', d['value'], d['ms_per_step'], d['verified'])
print('roofline', {k: d['roofline'][k] for k in ('achieved', 'frac', 'traffic', 'frac_counter', 'avg_launch_ms')})
print('cpu', {k: d['cpu_baseline'][k] for k in ('value', 'iqr', 'timed_frames', 'torch_threads')})
print('strong', d['configs3_strong']['value'], d['configs3_strong']['ms_per_step'])
print('other', {k: (v['value'], v['ms_per_step']) for k, v in d['other_workloads'].items()})
s = json.loads([l for l in open('gpurun_out/r05_ah/bench_soak_400.json') if l.startswith('{')][-1])
print('soak', s['value'], s['ms_per_step'], s['verified'])
PY
