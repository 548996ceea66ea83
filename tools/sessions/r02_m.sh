#!/bin/bash
O=gpurun_out/r02_m; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -k "batch32 or pair_pipeline or topk or golden or full_size or waymo or option_variants" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline > $O/bench_b32.json 2> $O/bench_b32.err; tail -2 $O/bench_b32.err; cut -c1-200 $O/bench_b32.json
for B in 1 4 8; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_b${B}.json 2> $O/bench_b${B}.err
  cut -c1-200 $O/bench_b${B}.json
done
python - <<'PY'
import json
for n in ('b1','b4','b8','b32'):
    d=json.loads([l for l in open(f'gpurun_out/r02_m/bench_{n}.json') if l.startswith('{')][-1])
    print(n, d['value'], d['ms_per_step'], d['roofline_dense']['dense_launches_ms'])
PY
