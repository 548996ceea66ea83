#!/bin/bash
# round 4 session zf: the driver's default bench line on the final tree (two-slot preflight in the configs[3] probe child)
mkdir -p gpurun_out
t0=$(date +%s)
timeout 300 python bench.py > gpurun_out/r04_zf_bench_default.json 2> gpurun_out/r04_zf_bench_default.err
echo "rc=$? wall=$(( $(date +%s) - t0 ))s"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_zf_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['execution'])
print(d.get('configs3_strong'))
print({k: v.get('value') for k, v in d.get('other_workloads', {}).items()} if isinstance(d.get('other_workloads'), dict) else d.get('other_workloads'))
print(d['roofline']['frac'], d['roofline_dense']['frac'], d['cpu_baseline']['value'])
PY
tail -3 gpurun_out/r04_zf_bench_default.err
