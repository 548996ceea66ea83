#!/bin/bash
# round 5 session d: lc chain as one captured graph (NeckAndHead), channels-last camera maps; lc bench A/B; the whole GPU suite
mkdir -p gpurun_out/r05_d
O=gpurun_out/r05_d
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -k "channels_last or neck_and_head" > $O/tests_lc.log 2>&1; echo "rc=$?" >> $O/tests_lc.log
tail -12 $O/tests_lc.log
timeout 500 python bench.py --workload lc --no-cpu-baseline --steps 10 > $O/bench_lc.json 2> $O/bench_lc.err
timeout 500 python bench.py --workload lc --no-cpu-baseline --steps 10 --graph off > $O/bench_lc_eager.json 2> $O/bench_lc_eager.err
FF3D_NECK_CAM_NHWC=0 timeout 500 python bench.py --workload lc --no-cpu-baseline --steps 10 --graph off > $O/bench_lc_eager_nchw.json 2> $O/bench_lc_eager_nchw.err
python - <<'PY'
import json
for n in ('lc', 'lc_eager', 'lc_eager_nchw'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/r05_d/bench_{n}.json') if l.startswith('{')][-1])
        print(n, d['value'], d['ms_per_step'], d['verified'], d['config']['execution'][:70])
    except Exception as e:
        print(n, 'no line', e)
PY
tail -3 $O/bench_lc.err
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_suite.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_suite.txt
tail -8 $O/pytest_gpu_suite.txt
