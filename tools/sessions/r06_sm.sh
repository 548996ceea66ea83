#!/bin/bash
# round 6, third session: the other workloads at one and two frames on the final tree (split-K pyramid convs, own projections at every row count): verified replays
O=$PWD/gpurun_out/r06_sm; mkdir -p $O
for wl in lc waymo l; do
  for b in 1 2; do
    timeout 400 python bench.py --workload $wl --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-strong-probe --no-other-workloads --no-companions 2>$O/err_${wl}_$b.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$wl batch $b: %.3f ms per step, %.1f frames/s, %s, verified %s' % (d['ms_per_step'], d['value'], d['config']['execution'][:48], json.dumps(d['verified'])[:90]))" | tee -a $O/small.txt
  done
done
