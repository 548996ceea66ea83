#!/bin/bash
# batch / execution-mode sweep of bench.py on one MI355X; results -> gpurun_out/sweep_*.json
mkdir -p gpurun_out
for B in 1 4 8 32; do
  python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/sweep_eager_b$B.json 2> gpurun_out/sweep_eager_b$B.err
  python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --graph > gpurun_out/sweep_graph_b$B.json 2> gpurun_out/sweep_graph_b$B.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/sweep_*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], 'frames/s', d['ms_per_step'], 'ms/step', 'msda frac', d['roofline']['frac'], d['roofline']['avg_launch_ms'])
    except Exception as e:
        print(f, 'FAILED', e, open(f.replace('.json','.err')).read()[-600:])
PY
