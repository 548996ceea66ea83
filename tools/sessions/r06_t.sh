#!/bin/bash
# round 6 session t: RoI sampler with the corner loads of NB items in flight (FF3D_ROI_NB = 1 .. 4): tests, microbench, step A/B
O=$PWD/gpurun_out/r06_t; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_round6_gpu.py -x -q -k "roi" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -n "Error\|assert \|passed\|failed" $O/tests.log | head -5
for sm in 1 0; do for v in 1 2 3 4 1 3; do SMALL=$sm FF3D_ROI_NB=$v timeout 200 python tools/experiments/exp_roi.py 2>&1 | grep -v amdgpu | sed "s/^/NB=$v /" >> $O/roi.txt; done; done
cat $O/roi.txt
b() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
b nb3_a; FF3D_ROI_NB=1 b nb1_a; b nb3_b; FF3D_ROI_NB=1 b nb1_b
python - <<'PY'
import json
for n in ('nb3_a', 'nb1_a', 'nb3_b', 'nb1_b'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r06_t/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'].get('bit_identical'))
    except Exception as e:
        print(n, 'no line', e)
PY
