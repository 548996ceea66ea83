#!/bin/bash
# round 6, third session: batches in flight at the 32-frame step on this tree (1 / 2 / 3 / 4 slots, same box, two alternations)
O=$PWD/gpurun_out/r06_sl; mkdir -p $O
for rep in 1 2; do
for sl in 2 3 1 4; do
  timeout 400 python bench.py --slots $sl --steps 30 --warmup 4 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('slots $sl: %.3f ms per step, %.1f frames/s, verified %s' % (d['ms_per_step'], d['value'], d['verified'].get('bit_identical')))" | tee -a $O/slots.txt
done
done
