#!/bin/bash
O=$PWD/gpurun_out/r02_af; mkdir -p $O
export TMPDIR=/tmp
for v in 0 4; do FF3D_SPLITMM_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_v$v.json 2> $O/bench_v$v.err; python - <<PY
import json
d=json.loads([l for l in open('$O/bench_v$v.json') if l.startswith('{')][-1])
print('variant $v', d['value'], d['roofline_dense']['dense_launches_ms'])
PY
done
