#!/bin/bash
# round 4 session m: MSDA gather with the 16 corner loads of a level issued together (P == 4): parity tests, A/B, default bench
O=$PWD/gpurun_out/r04_m; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_head_gpu.py tests/test_bench_shape_gpu.py -x -q -m gpu -k "msda or head or decoder or batch4" > $O/pytest_msda.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest_msda.log | cut -c1-300
for w in 1 0 1 0; do FF3D_MSDA_PT4=$w timeout 120 python tools/experiments/exp_msda.py 2>&1 | grep "PT4" | tee -a $O/msda_pt4_ab.txt; done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_m/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'configs3', d['configs3_strong'].get('value'), 'roofline', d['roofline']['frac'], d['roofline']['frac_counter'], d['roofline']['avg_launch_ms'])
PY
FF3D_MSDA_PT4=0 timeout 600 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads > $O/bench_pt4_off.json 2> $O/bench_pt4_off.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_m/bench_pt4_off.json').read().strip().splitlines()[-1])
print('PT4 off:', d['value'], d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
