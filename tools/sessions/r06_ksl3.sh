#!/bin/bash
# round 6, third session: K-sliced fc2 also at four frames (2 400 rows)?  pipelined 4-frame step, threshold 2 560 vs 1 536, three alternations
O=$PWD/gpurun_out/r06_ksl3; mkdir -p $O
for rep in 1 2 3; do
for mr in 2560 1536; do
FF3D_LIN_LN_KSLICES_MAX_ROWS=$mr timeout 300 python bench.py --batch 4 --steps 300 --warmup 30 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('max rows $mr batch 4 pipelined: %.4f ms per step, %.1f frames/s, verified %s' % (d['ms_per_step'], d['value'], d['verified'].get('bit_identical')))" | tee -a $O/ab.txt
done
done
