#!/bin/bash
O=gpurun_out/r02_u; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 800 -k "conv3x3 or split_out or nhwc_pair or golden or any_magnitude or halo or pair_pipeline" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log | cut -c1-250
for m in pair none; do
  FF3D_TR=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_b32_$m.json 2> $O/bench_b32_$m.err
  FF3D_TR=$m timeout 300 python tools/bench_neck.py 32 > $O/neck_$m.json 2> $O/neck_$m.err
done
python - <<'PY'
import json
for n in ('pair','none'):
    d=json.loads([l for l in open(f'gpurun_out/r02_u/bench_b32_{n}.json') if l.startswith('{')][-1])
    print(n, d['value'], d['ms_per_step'], d['roofline_dense']['dense_launches_ms'])
    print(open(f'gpurun_out/r02_u/neck_{n}.json').read()[-250:])
PY
