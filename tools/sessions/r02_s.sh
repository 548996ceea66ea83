#!/bin/bash
O=$PWD/gpurun_out/r02_s; mkdir -p $O
FF3D_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/rccl1.json 2> $O/rccl1.err
python -c "
import json; d=json.loads([l for l in open('$O/rccl1.json') if l.startswith('{')][-1]); print('rccl 1-rank', d['value'], d.get('configs3_strong'))"
FF3D_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/gloo2.json 2> $O/gloo2.err
python -c "
import json; d=json.loads([l for l in open('$O/gloo2.json') if l.startswith('{')][-1]); print('gloo 2-rank', d['value'], d.get('configs3_strong'))"
FF3D_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/gloo2_b8.json 2> $O/gloo2_b8.err
python -c "
import json; d=json.loads([l for l in open('$O/gloo2_b8.json') if l.startswith('{')][-1]); print('gloo 2-rank b8', d['value'], d.get('configs3_strong'))"
