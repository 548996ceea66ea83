#!/bin/bash
# round 4 session g: wide NMS kernel without the LDS histogram - A/B, rocprof duration of the kernel, whole GPU suite, smoke, default bench
O=$PWD/gpurun_out/r04_g; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
for w in 1 0 1 0; do echo "FF3D_NMS_WIDE=$w:" | tee -a $O/nms_wide_ab.txt; FF3D_NMS_WIDE=$w timeout 120 python tools/experiments/exp_nms.py 2>&1 | grep "B=" | tee -a $O/nms_wide_ab.txt; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_nms -o r -- python $R/tools/experiments/exp_nms.py > $O/nms_under_rocprof.txt 2> $O/rocprof_nms.err )
DB=$(find $O/prof_nms -name '*_results.db' | head -1)
python tools/rocprof_summary.py $DB 12 > $O/nms_kernel_stats.txt 2>&1; find $O/prof_nms -name '*.db' -delete
grep -i "nms\|fill" $O/nms_kernel_stats.txt | head -6 | cut -c1-200
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -3 $O/pytest_all.log | cut -c1-300
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_g/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['execution'][:60], 'configs3', d['configs3_strong'].get('value'), d['configs3_strong'].get('projected_speedup_8_vs_1'), {k:(v.get('value'),v.get('error')) for k,v in d['other_workloads'].items()})
print('roofline', d['roofline']['frac'], d['roofline']['frac_counter'], 'dense', d['roofline_dense']['frac'], 'cpu', d['cpu_baseline']['value'])
PY
