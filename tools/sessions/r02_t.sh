#!/bin/bash
O=gpurun_out/r02_t; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q --timeout 1500 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log | cut -c1-250
for B in 32 4; do
  timeout 300 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b${B}.json 2> $O/bench_b${B}.err
  FF3D_SPLITMM_NO_TR=1 timeout 300 python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b${B}_notr.json 2> $O/bench_b${B}_notr.err
done
python - <<'PY'
import json
for n in ('b32','b32_notr','b4','b4_notr'):
    d=json.loads([l for l in open(f'gpurun_out/r02_t/bench_{n}.json') if l.startswith('{')][-1])
    print(n, d['value'], d['ms_per_step'], d['roofline_dense']['dense_launches_ms'])
PY
